"""Stride-2 3x3x3 convolution (TF 'same' padding on even dims: one zero plane BEHIND the volume, out[o] = sum_t W[t] x[2o + t];
reference isensee2017.py:51 create_convolution_block(..., strides=(2, 2, 2))) on the parity kernels of the up-sampling path.

The input voxels 2o + t a stride-2 output reads are, per axis, parity 0 of block o (t = 0), parity 1 of block o (t = 1) and parity 0 of
block o + 1 (t = 2) - the gather the "up-backward" launch (k_conv_fwd_ws MODE 2, fmri_conv3d_upcat_dgrad) performs over the space-to-depth
view of a full-resolution tensor: 8 parity classes x 2x2x2 block offsets, of which a 3-tap axis uses 3 of 4.  So
    forward          = fmri_conv3d_upcat_dgrad(x, w_s2_fwd)          64 slots, 27 of them non-zero: 2.4x the ideal MACs instead of the 8x of
                                                                     "stride-1 conv, keep every second voxel", no full-resolution temporary
    input gradient   = fmri_conv3d_upcat_fwd(dy, w_s2_dgrad)         (the scatter twin, MODE 1)
    weight gradient  = fmri_conv3d_upcat_wgrad(dy, x) -> 64 slot gradients, 27 of them read back
with the slot images built here.  Per axis: tap t -> (parity p, slot u) of w_up_dgrad's [p][u] indexing (input voxel 2 (o + u - p) + p):
t = 0 -> (0, 0), 1 -> (1, 1), 2 -> (0, 1); and -> (parity p, combined tap t') of w_up_fwd's (input voxel 2g + p receives from output
g + t' - 1 + p): t = 0 -> (0, 1), 1 -> (1, 0), 2 -> (0, 0)."""
import numpy as np
import torch

_GATHER = {0: (0, 0), 1: (1, 1), 2: (0, 1)}          # t -> (p, u)   forward image (w_up_dgrad layout [8 p][8 u][C_low][C_fine])
_SCATTER = {0: (0, 1), 1: (1, 0), 2: (0, 0)}         # t -> (p, t')  input-gradient image (w_up_fwd layout [8 p][8 t'][C_fine][C_low])


def _tables(m):
    P, S = np.zeros(27, np.int64), np.zeros(27, np.int64)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                (pd, sd), (ph, sh), (pw, sw) = m[kd], m[kh], m[kw]
                t = (kd * 3 + kh) * 3 + kw
                P[t], S[t] = pd * 4 + ph * 2 + pw, sd * 4 + sh * 2 + sw
    return P, S


class StridedParity:
    """index tables on `device` + the three data movements between the 27-tap filter and the 64-slot images"""

    def __init__(self, device):
        self.gp, self.gu = (torch.from_numpy(a).to(device) for a in _tables(_GATHER))
        self.sp, self.st = (torch.from_numpy(a).to(device) for a in _tables(_SCATTER))

    def pack(self, w27, fwd_img, dgrad_img):
        """w27 [27][Cout][Cin] fp32 -> fwd_img [8][8][Cout][Cin] (as w_up_dgrad: C_low = Cout, C_fine = Cin), dgrad_img [8][8][Cin][Cout]
        (as w_up_fwd); the 37 unused slots of either must be zero (they are never written here: allocate the images zeroed)"""
        fwd_img[self.gp, self.gu] = w27.to(fwd_img.dtype)
        if dgrad_img is not None:
            dgrad_img[self.sp, self.st] = w27.transpose(1, 2).to(dgrad_img.dtype)

    def unpack_wgrad(self, dwc, cout, cin):
        """dwc: the slot gradients fmri_conv3d_upcat_wgrad leaves for (low-res source = dy [.., Cout], full-resolution 'gradient' = x [.., Cin]),
        [8 p][8 t'][Cin][Cout] fp32 -> dW [27][Cout][Cin]"""
        return dwc[:64 * cin * cout].view(8, 8, cin, cout)[self.sp, self.st].transpose(1, 2)
