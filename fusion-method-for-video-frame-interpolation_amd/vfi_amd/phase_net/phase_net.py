"""PhaseNet -- mirror of reference src/phase_net/phase_net.py (the variant the fused path uses:
`PhaseNet(pyr, device, num_img=2)`, two-call protocol `normalize_vals(vals)` then `forward(vals)`).

Execution on the MI355X (all arithmetic in libvfi_hip.so):
  * block = [conv k, BN(eval), ELU, conv k, ELU] -> 64 features; [1x1 conv, tanh] -> prediction
    (phase_net.py:190-200): three fp32-MFMA conv launches, BN folded into the first conv's weights,
    ELU / tanh in the conv epilogues; k = 1 for blocks 0-2, 3 (reflect) for 3-7 (phase_net.py:30-35);
  * the block input `cat(feature_r, phase, amp, prediction_r)` (phase_net.py:141) is never
    concatenated: each level owns one buffer laid out [feature 64 | prediction P | phase 8 | amp 8]
    (the first conv's input channels are permuted to match at pack time); the previous block writes
    feature and prediction into ONE tensor, so a single bilinear-resize launch fills the first 64+P
    channels, and normalize_vals writes phase/pi and amp/max straight into the last 16;
  * the per-level blend + de-normalisation (phase_net.py:155-168, :80-105) is one launch per level.
Maxima are returned/kept per call on the module (as the reference does, phase_net.py:53,59,70).
"""
import math

import torch

from .. import _lib, ops
from ..nn_util import BatchNormParams, ConvParams, Indexed, PackedModule
from ..values import DecompValues, NormalizedValues


class PhaseNetBlock(torch.nn.Module):
    """Parameter holder with the reference's key names (phase_net.py:179-200)."""

    def __init__(self, c_in, c_out, pred_out, kernel_size, device=None, dropout=0.5):
        super().__init__()
        k = kernel_size[0]
        self.feature_map = Indexed({0: ConvParams(c_in, c_out, k), 1: BatchNormParams(c_out),
                                    3: ConvParams(c_out, c_out, k)})
        self.prediction_map = Indexed({0: ConvParams(c_out, pred_out, 1)})


class PhaseNet(PackedModule):
    def __init__(self, pyr, device, num_img=2):
        super().__init__()
        if num_img != 2:
            raise NotImplementedError("vfi_amd.PhaseNet implements the two-frame network of the fused path")
        self.pyr = pyr
        self.device = torch.device(device)
        self.num_img = num_img
        self.eps = 1e-8
        blocks = [PhaseNetBlock(num_img, 64, 1, (1, 1)),
                  PhaseNetBlock(64 + 1 + 8 * num_img, 64, 8, (1, 1)),
                  PhaseNetBlock(64 + 8 + 8 * num_img, 64, 8, (1, 1))]
        blocks += [PhaseNetBlock(64 + 8 + 8 * num_img, 64, 8, (3, 3)) for _ in range(5)]
        self.layers = torch.nn.ModuleList(blocks)
        self.max_amplitudes = None
        self.max_low_level = None
        self.train(False)
        self.to(self.device)

    # -- weights ------------------------------------------------------------------------------------
    def _build_packed(self):
        out = []
        for i, blk in enumerate(self.layers):
            w = blk.feature_map[0].weight
            if i >= 1:
                # reference channel order [feature 64 | phase 8 | amp 8 | pred P] -> ours
                # [feature 64 | pred P | phase 8 | amp 8]
                p = w.shape[1] - 80
                perm = list(range(64)) + list(range(80, 80 + p)) + list(range(64, 80))
                w = w[:, perm]
            c1 = ops.PackedConv(w, blk.feature_map[0].bias, bn=blk.feature_map[1].fold_args())
            out.append((c1, self.pack(blk.feature_map[3]), self.pack(blk.prediction_map[0])))
        return out

    # -- normalisation (phase_net.py:42-78) -------------------------------------------------------------
    def normalize_vals(self, vals, concat=None, amp_max=None):
        """phase_net.py:42-78.  `concat` (from Pyramid.filter(concat_frames=2, phase_scale=1/pi)): the block-input
        buffers that already hold phase/pi and the raw amplitudes -- then only the maxima are computed and
        the amplitudes / low level are normalised in place (no copies)."""
        if concat is not None:
            return self._normalize_in_place(vals, concat, amp_max)
        nlev = len(vals.phase)
        b = vals.amplitude[0].shape[0]
        maxes, concat, phases, amps = [], [], [], []
        for idx in range(nlev):
            amp, ph = vals.amplitude[idx].contiguous(), vals.phase[idx].contiguous()
            mx = ops.batch_max(amp, self.eps)                                   # :55
            maxes.append(mx)
            p_prev = 1 if idx == 0 else 8
            _, c, h, w = amp.shape
            buf = ops.new((b, 64 + p_prev + 2 * c, h, w), amp)
            pv = buf[:, 64 + p_prev:64 + p_prev + c]
            av = buf[:, 64 + p_prev + c:64 + p_prev + 2 * c]
            ops.affine_slice(ph, pv, None, 1.0 / math.pi)                        # :64
            ops.affine_slice(amp, av, mx, 1.0)                                   # :61
            concat.append(buf); phases.append(pv); amps.append(av)
        low_in = vals.low_level.contiguous()
        self.max_amplitudes = maxes
        self.max_low_level = ops.batch_max(low_in, self.eps)                     # :69
        low = ops.affine_slice(low_in, torch.empty_like(low_in), self.max_low_level, 1.0)   # :70
        out = NormalizedValues(vals.high_level, phases, amps, low)
        out.concat = concat
        return out

    def _normalize_in_place(self, vals, concat, amp_max=None):
        """amp_max (levels coarsest first, colours): maxima + eps already reduced by Pyramid.filter(amp_max_eps=...)."""
        maxes = []
        for idx, amp in enumerate(vals.amplitude):          # views into concat[idx]
            mx = amp_max[idx] if amp_max is not None else ops.batch_max(amp, self.eps)
            maxes.append(mx)
            ops.affine_slice(amp, amp, mx, 1.0)
        self.max_amplitudes = maxes
        self.max_low_level = ops.batch_max(vals.low_level, self.eps)
        ops.affine_slice(vals.low_level, vals.low_level, self.max_low_level, 1.0)
        out = NormalizedValues(vals.high_level, list(vals.phase), list(vals.amplitude), vals.low_level)
        out.concat = concat
        return out

    def reverse_normalize(self, vals, m):
        """phase_net.py:80-105 on already-blended outputs (generic path; forward() fuses this)."""
        phases = [ops.affine_slice(p.contiguous(), torch.empty_like(p), None, math.pi) for p in vals.phase]
        amps = []
        for i in range(m):
            a = vals.amplitude[i]
            nb = self.pyr.nbands
            a4 = a.reshape(a.shape[0] // nb, nb, a.shape[2], a.shape[3]).contiguous()
            inv = 1.0 / self.max_amplitudes[i]
            amps.append(ops.affine_slice(a4, torch.empty_like(a4), inv, 1.0).reshape(a.shape))
        for _ in range(self.pyr.height - 2 - m):
            phases.append(0); amps.append(0)
        low = ops.affine_slice(vals.low_level.contiguous(), torch.empty_like(vals.low_level),
                               1.0 / self.max_low_level, 1.0)
        return DecompValues(vals.high_level, phases[::-1], amps[::-1], low)

    # -- forward (phase_net.py:107-177) ------------------------------------------------------------------
    def forward(self, vals, m=None):
        if m is None:
            m = self.pyr.height - 2
        if self.max_amplitudes is None:
            raise RuntimeError("call normalize_vals(vals) before forward(vals) (phase_net.py two-call protocol)")
        packed = self.packed()
        low_in = vals.low_level.contiguous()
        b, _, hl, wl = low_in.shape
        stream = _lib.stream_ptr()

        def block(i, x, fp, prev=None, head=True):
            c1, c2, cp = packed[i]
            mode = "reflect" if c1.ks == 3 else "zeros"
            if prev is not None:     # 3x3 block: the resize of (feature | prediction) is done by the conv's tile loader
                t = ops.conv2d_resized_prefix(x, prev, c1, mode, "elu")
            else:
                t = ops.conv2d(x, c1, mode, "elu")
            ops.conv2d(t, c2, mode, "elu", out=fp[:, :64])
            if head:                 # (the band levels' prediction map comes out of vfi_phasenet_predict with their outputs)
                ops.conv2d(fp[:, :64], cp, "zeros", "tanh", out=fp[:, 64:])
            return fp

        fp = block(0, low_in, ops.new((b, 65, hl, wl), low_in))                       # :113
        low = ops.new((b, 1, hl, wl), low_in)
        _lib.call("vfi_phasenet_emit_low", fp[:, 64:].data_ptr(), fp.stride(0), low_in.data_ptr(), low_in.stride(0),
                  self.max_low_level.data_ptr(), low.data_ptr(), b, hl * wl, stream)     # :115-116 + :96-98
        hs = vals.high_level.shape
        high = torch.zeros((hs[0], 1, hs[2], hs[3]), dtype=torch.float32, device=low_in.device)   # :127-128

        concat = getattr(vals, "concat", None)
        phases, amps = [], []
        for idx in range(m):
            p_prev = fp.shape[1] - 64
            ph, am = vals.phase[idx], vals.amplitude[idx]
            _, c, h, w = ph.shape
            if concat is not None:
                x = concat[idx]
            else:  # values not produced by normalize_vals: fill the block input buffer here
                x = ops.new((b, 64 + p_prev + 2 * c, h, w), low_in)
                ops.affine_slice(ph.contiguous(), x[:, 64 + p_prev:64 + p_prev + c])
                ops.affine_slice(am.contiguous(), x[:, 64 + p_prev + c:])
            i = idx + 1 if idx + 1 < len(self.layers) - 1 else len(self.layers) - 1        # :148
            fused = packed[i][0].ks == 3 and (64 + p_prev) % 8 == 0
            if not fused:
                ops.resize_bilinear(fp, (h, w), align_corners=False, out=x[:, :64 + p_prev])   # :138-141
            fp = block(i, x, ops.new((b, 72, h, w), low_in), prev=fp if fused else None, head=False)
            amp_in = x[:, 64 + p_prev + c:]
            p_out, a_out = ops.new((b * 4, 1, h, w), low_in), ops.new((b * 4, 1, h, w), low_in)
            # prediction map (1x1, tanh) + this level's outputs in one pass over the 64 feature channels (:149-168)
            cp = packed[i][2]
            if cp.cout != 8 or cp.ks != 1:
                raise RuntimeError("PhaseNet: a band level's prediction map must be a 1x1 layer with 8 outputs (phase_net.py:30-35)")
            _lib.call("vfi_phasenet_predict", fp.data_ptr(), fp.stride(0), cp.packed.data_ptr(), cp.bias.data_ptr(), amp_in.data_ptr(),
                      x.stride(0), self.max_amplitudes[idx].data_ptr(), fp[:, 64:].data_ptr(), fp.stride(0), p_out.data_ptr(),
                      a_out.data_ptr(), b, 64, h, w, stream,
                      work=("byte", 4.0 * b * (64 + 8 + 8 + 8) * h * w, "phasenet_predict_kernel") if _lib.PROFILE is not None else None)
            phases.append(p_out); amps.append(a_out)
        for _ in range(self.pyr.height - 2 - m):                                           # :91-93
            phases.append(0); amps.append(0)
        return DecompValues(high, phases[::-1], amps[::-1], low)
