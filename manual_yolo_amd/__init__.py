"""MI355X-native YOLOv8 detect + classify inference path (drop-in for the reference's
``from ultralytics import YOLO`` at ``detect.py:9``, ``pipe.py``, ``yolo.py``, ``class.py``).

    from manual_yolo_amd import YOLO
    model = YOLO("poker_model.pt")          # or a .safetensors bundle
    results = model(frame)[0]               # reference detect.py:541

Compute runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/miyolo.h``
(``manual_yolo_amd/csrc``); PyTorch is used for device memory, streams and
``torch.distributed`` only.  There is no CPU fallback: without the built extension and a
GPU, constructing an engine raises.
"""
from .results import Boxes, Probs, Results  # noqa: F401


def __getattr__(name):
    if name == "YOLO":
        from .model import YOLO
        return YOLO
    raise AttributeError(name)


__all__ = ["YOLO", "Results", "Boxes", "Probs"]
