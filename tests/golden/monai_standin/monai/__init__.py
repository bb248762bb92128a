"""TEST INFRASTRUCTURE ONLY — a torch-only stand-in for the handful of `monai==1.1.0` names that
/root/reference/model/dim3/swin_unetr.py:24-27 imports (monai is not installed in this image and there is no
network).  Written from the published MONAI 1.1.0 behaviour, NOT pinned against a MONAI wheel:
everything produced through these blocks is "parity unpinned" for the MONAI part; the in-tree Swin transformer
code of the reference runs unmodified on top of it.  Used only by tests/golden/make_golden_swin.py.
"""
