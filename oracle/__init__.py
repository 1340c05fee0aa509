"""CPU oracle for the YOLOv8 detect + classify inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``manual_yolo_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / the timed CPU baseline.

What it restates (the arithmetic lives in third-party packages that are NOT
vendored under /root/reference and are not installed in this image):

* ``ultralytics==8.3.176``  (reference ``requirements.txt:95``)  - nn modules
  (Conv/C2f/Bottleneck/SPPF/Detect/DFL/Classify), ``fuse_conv_and_bn``,
  ``non_max_suppression``, ``scale_boxes``, ``LetterBox``
* ``torchvision==0.23.0``   (reference ``requirements.txt:89``)  - ``ops.nms``
  (CPU kernel semantics) and the classify transforms
* ``torch==2.8.0``          (reference ``requirements.txt:88``)  - ATen CPU ops;
  the oracle runs on the torch that IS installed here (same ATen CPU kernels).

Call sites the oracle stands in for: reference ``detect.py:541``
(``model(frame)[0]``), ``detect.py:121`` (``rank_model(crop)[0]``),
``pipe.py:179`` (``predict(imgsz=1280, conf=0.35)``), ``yolo.py:361``.

Parity status
-------------
* classification: PINNED by the reference's own artefacts
  (``runs/rank_classifier/results.csv:21-22`` and ``confusion_matrix.png``:
  best.pt 63/67 with errors 10->Q, 8->3, 9->6, Q->6; last.pt 61/67) -
  see ``tests/test_oracle_kat.py``.
* detection + NMS: PARITY UNPINNED - ``poker_model.pt`` is absent
  (reference ``.MISSING_LARGE_BLOBS:3``) and no reference test or output holds a
  detection number.  The architecture is pinned only by the published
  yolov8m parameter count (25 902 640 @ nc=80).
"""
