"""Host-side mirror of the reference `fetal_net` package for the hot path only:
model builders, metrics tokens, training driver, sliding-window prediction.  The compute runs in libfmri_hip.so
(include/fmri_hip.h) on MI355X; there is no CPU implementation of the network in this package."""
from . import metrics  # noqa: F401
