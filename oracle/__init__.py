"""CPU oracle package — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(fetal-mri-segmentation_amd/) never does and fails loudly when its HIP library is missing.
"""
