"""ORACLE — test infrastructure, NOT product code.

CPU restatement (torch fp32 + plain C) of the reference's scan -> BEV -> mask
forward/backward path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; ``mask_bev_amd``
never does.

Pinning status:
  * in-repo arithmetic (swin.py, mask2former_head.py) is pinned by the golden
    vectors under tests/golden/ produced by tests/golden/make_golden.py, which
    imports the reference's own files in the build container;
  * arithmetic that lives in the un-vendored third-party packages
    (mmcv 2.0.0, mmdet 3.0.0, mmdet3d 1.1.0) is restated from their published
    algorithms — PARITY UNPINNED, no reference test holds values for it.
"""
