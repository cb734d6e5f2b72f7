#!/bin/bash
# round 6: the option switches still pass the suites that exercise them
echo "== FMRI_F32_MFMA=0 (fp32 on the VALU kernels)"; FMRI_F32_MFMA=0 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_gpu_val_dice.py -q -p no:cacheprovider 2>&1 | tail -3
echo "== FMRI_UPW_KD=2 / FMRI_WGRAD_KD_BLK=64"; FMRI_UPW_KD=2 FMRI_WGRAD_KD_BLK=64 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_engine.py -q -p no:cacheprovider -k "upcat or parity or wgrad or weight or engine or depth4" 2>&1 | tail -3
