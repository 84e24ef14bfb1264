"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain CPU restatement (torch fp32 functional ops + a plain-C
cross-check) of the reference's per-pixel restoration generators.  It exists so
that the HIP path can be checked for parity.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``etss-07-imageenhance_amd`` / ``mon_amd``) never does
and fails loudly when its HIP library is missing.

Parity pin: the restatement is checked against outputs of the reference's own
importable model files (``src/mon_extra/vision/enhance/llie/{zero_dce,
zero_dce++,llunet++}/model.py``) captured in this container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``.
Where a path has no importable reference file (U-Net, RetinexNet, SSIM) the
header of the corresponding oracle module says "parity unpinned" or names the
pin it has.
"""
