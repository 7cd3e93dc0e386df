"""CPU oracle for the ifcb_classifier hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``ifcb_classifier_amd/`` imports it; the product path fails loudly
when the HIP library is missing instead of falling back to this code.

PARITY UNPINNED: the reference (WHOIGit/ifcb_classifier v0.3.1) ships no tests,
golden vectors or fixtures for this path, and its arithmetic lives in third-party packages that are
not under /root/reference: torch==1.7.1 (ATen primitives; torch 2.10 CPU is used here), torchvision==0.8.2
(model graphs; restated in ``tv_models.py``, pinned only by its published parameter counts
27,161,264 / 11,689,512 and state_dict key list), pillow==8.4.0 (``ImagingResample``; restated in
``pil_resize.py`` and pinned against Pillow itself, which IS installed), pytorch-lightning==1.3.8
(loop order; restated in ``step.py``).
"""
