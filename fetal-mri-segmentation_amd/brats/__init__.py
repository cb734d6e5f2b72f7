"""Alias kept because the reference README and prod/ scripts still import `brats.*` (reference fetal/predict2.py:9,12)."""
