"""Audio stem on the HIP GEMM: the wav2vec2-style strided Conv1d stack and the grouped
convolutional positional encoder of adapter/audio.py:46-84,254-311 as channels-last GEMMs over *strided views* of the
activations -- no im2col buffers, no [B,C,T] <-> [B,T,C] transposes, no MIOpen.

Layout: every feature map is a flat channels-last matrix ``[B * T_slot (+2 slack rows), C]`` where ``T_slot`` is the
sample's slot count (waveform length / cumulative stride).  Because the waveform length is padded to a multiple of
320 (= total stride), sample b's frames always start at row ``b * T_slot`` and a stride-2 convolution is a GEMM whose A
operand is the same buffer viewed with row stride ``2 * C``:

    y[r] = W[:, :, 0:2] . (x[2r] | x[2r+1])  +  W[:, :, 2] . x[2r+2]            (k = 3; k = 2 drops the second term)

The last slot(s) of each sample hold values computed from the neighbouring sample ("garbage rows"); valid outputs never
read them (conv arithmetic), the adapter slices them off at the end, and autograd therefore feeds them zero gradients,
so they contribute nothing to any weight gradient."""
import torch
import torch.nn.functional as F

from . import hip, ops

TOTAL_STRIDE = 320  # 5 * 2**6


def _as_rows(buf, rows, cols, row_stride, offset_elems):
    return torch.as_strided(buf, (rows, cols), (row_stride, 1), buf.storage_offset() + offset_elems)


class StridedConv1dFn(torch.autograd.Function):
    """Conv1d(Cin -> Cout, kernel k in {2, 3}, stride 2, no bias), channels-last, flat rows.
    x: [2*R + 2, Cin] (R output rows + slack)  ->  y: [R + 2, Cout] (2 zero slack rows)."""

    @staticmethod
    def forward(ctx, x, weight):
        Cout, Cin, k = weight.shape
        R = (x.shape[0] - 2) // 2
        w01 = weight[:, :, :2].permute(0, 2, 1).reshape(Cout, 2 * Cin).contiguous()
        y = torch.empty(R + 2, Cout, dtype=x.dtype, device=x.device)
        y[R:].zero_()
        a01 = _as_rows(x, R, 2 * Cin, 2 * Cin, 0)
        hip.gemm_nt(a01, [w01], out=y[:R])
        w2 = None
        if k == 3:
            w2 = weight[:, :, 2].contiguous()
            a2 = _as_rows(x, R, Cin, 2 * Cin, 2 * Cin)
            hip.gemm_nt(a2, [w2], out=y[:R], epilogue=hip.EPI_RESID, resid=y[:R])
        ctx.save_for_backward(x, w01, w2)
        ctx.dims = (Cout, Cin, k, R)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w01, w2 = ctx.saved_tensors
        Cout, Cin, k, R = ctx.dims
        dy = dy.contiguous()
        dyv = dy[:R]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            dx[2 * R:].zero_()
            hip.gemm_nt(dyv, [hip.transpose(w01)], out=_as_rows(dx, R, 2 * Cin, 2 * Cin, 0))
            if k == 3:
                tgt = _as_rows(dx, R, Cin, 2 * Cin, 2 * Cin)
                hip.gemm_nt(dyv, [hip.transpose(w2)], out=tgt, epilogue=hip.EPI_RESID, resid=tgt)
        if ctx.needs_input_grad[1]:
            # dW = dy^T x straight from the strided activation views (transpose-read GEMM, no transposed copies)
            dw01 = ops.wgrad(dyv, _as_rows(x, R, 2 * Cin, 2 * Cin, 0))  # [Cout, 2*Cin] as (tap, cin)
            dw = torch.empty(Cout, Cin, k, dtype=x.dtype, device=x.device)
            dw[:, :, :2] = dw01.view(Cout, 2, Cin).permute(0, 2, 1)
            if k == 3:
                dw[:, :, 2] = ops.wgrad(dyv, _as_rows(x, R, Cin, 2 * Cin, 2 * Cin))
        return dx, dw


class Conv1LnGeluFn(torch.autograd.Function):
    """Block 0 of the feature extractor -- Conv1d(1 -> C, k = 10, stride 5) -> LayerNorm(C) -> GELU -- as ONE kernel each way, straight from
    the flat waveform (csrc/audio.hip): the 2.1 GB convolution output of the headline batch is neither written nor kept; backward
    recomputes each row from its ten samples and accumulates the four parameter gradients in registers.  wav gets no gradient.
    Returns [rows + slack, C] with `slack` zero rows at the end (the next block's patch view reads past the last row): appending them with
    torch.cat copied the 2.1 GB matrix of the headline batch once more (1.4 ms)."""

    @staticmethod
    def forward(ctx, wav_flat, rows, stride, weight, bias, ln_w, ln_b, eps, slack=0):
        w0 = weight.reshape(weight.shape[0], 10).contiguous()
        y, mean, rstd = hip.audio_conv1_ln_gelu_fwd(wav_flat, stride, w0, bias, ln_w, ln_b, rows, eps, slack_rows=slack)
        ctx.save_for_backward(wav_flat, w0, bias, ln_w, ln_b, mean, rstd)
        ctx.stride, ctx.wshape, ctx.rows = stride, tuple(weight.shape), rows
        return y

    @staticmethod
    def backward(ctx, dy):
        wav_flat, w0, bias, ln_w, ln_b, mean, rstd = ctx.saved_tensors
        dw0, db0, dlw, dlb = hip.audio_conv1_ln_gelu_bwd(dy[:ctx.rows].contiguous(), wav_flat, ctx.stride, w0, bias, ln_w, ln_b, mean, rstd)
        return None, None, None, dw0.view(ctx.wshape), db0, dlw, dlb, None, None


FUSED_CONV1 = __import__("os").environ.get("ONEPEACE_FUSED_CONV1", "1") != "0"


def _first_layer_rows(wav_flat, rows):
    """im2col of Conv1d(1 -> C, k = 10, stride 5) as a strided view of the flat (padded) waveform, K padded to 64."""
    a = torch.as_strided(wav_flat, (rows, 10), (5, 1), wav_flat.storage_offset())
    out = torch.zeros(rows, 64, dtype=wav_flat.dtype, device=wav_flat.device)
    out[:, :10] = a
    return out


def feature_extractor(src_audios, conv_blocks):
    """ConvFeatureExtractionModel.forward (adapter/audio.py:254-311) on bf16 device tensors.
    src_audios [B, T_wav]; conv_blocks: the module list (each: Sequential(conv, dropout, Sequential(_, LayerNorm, _), GELU)).
    Returns channels-last features [B, T_frames, C]."""
    B, T = src_audios.shape
    Tp = (T + TOTAL_STRIDE - 1) // TOTAL_STRIDE * TOTAL_STRIDE
    valid = T
    slots = Tp // 5
    wav = torch.zeros(B * Tp + 16, dtype=src_audios.dtype, device=src_audios.device)
    wav[: B * Tp].view(B, Tp)[:, :T] = src_audios
    conv0, ln0 = conv_blocks[0][0], conv_blocks[0][2][1]
    rows = B * slots
    C0 = conv0.weight.shape[0]
    if (FUSED_CONV1 and conv0.weight.shape[2] == 10 and conv0.stride[0] == 5 and C0 <= 512 and C0 % 8 == 0 and wav.dtype == torch.bfloat16):
        x = Conv1LnGeluFn.apply(wav, rows, 5, conv0.weight, conv0.bias, ln0.weight, ln0.bias, ln0.eps, 2)  # (+ 2 slack rows)
    else:
        with torch.no_grad():
            a0 = _first_layer_rows(wav, rows)
        w0 = F.pad(conv0.weight.reshape(conv0.weight.shape[0], 10), (0, 54))
        x = ops.linear(a0, w0, conv0.bias)
        x = ops.layer_norm(x, ln0.weight, ln0.bias, ln0.eps, gelu=True)
        x = torch.cat([x, x.new_zeros(2, x.shape[1])], dim=0)  # slack rows
    valid = (valid - 10) // 5 + 1
    for block in conv_blocks[1:]:
        conv, ln = block[0], block[2][1]
        k = conv.weight.shape[2]
        assert conv.stride[0] == 2 and k in (2, 3) and conv.bias is None
        x = StridedConv1dFn.apply(x, conv.weight)
        x = ops.layer_norm(x, ln.weight, ln.bias, ln.eps, gelu=True)
        slots //= 2
        valid = (valid - k) // 2 + 1
    C = x.shape[1]
    return x[: B * slots].view(B, slots, C)[:, :valid]


class GroupedConv1dSameFn(torch.autograd.Function):
    """Conv1d(C -> C, odd kernel k, padding k//2, groups G, with bias) on channels-last [B, T, C]
    (the positional encoder blocks of adapter/audio.py:57-84).  One GEMM per group over a group-major, zero-padded copy
    whose rows are read as overlapping k*cg-wide patches (row stride cg)."""

    @staticmethod
    def _run(x, wg, bias_g, B, T, G, cg, k):
        """x [B, T, C] -> conv output rows [G, B*Ts, cg] (Ts = T + k - 1); wg [G, cg, Kp]."""
        P = k // 2
        Ts = T + 2 * P
        Kp = wg.shape[2]
        xg = torch.zeros(G, B * Ts + k + 1, cg, dtype=x.dtype, device=x.device)
        xg[:, : B * Ts].view(G, B, Ts, cg)[:, :, P:P + T] = x.view(B, T, G, cg).permute(2, 0, 1, 3)
        y = torch.empty(G, B * Ts, cg, dtype=x.dtype, device=x.device)
        if cg % 8 == 0 and xg.stride(0) % 8 == 0:  # the G products as ONE launch (each alone fills half the chip: 16 launches per layer before)
            hip.gemm_nt_batched(xg, wg, bias_g.contiguous() if bias_g is not None else None, y, B * Ts, Kp)
            return xg, y
        for g in range(G):
            a = _as_rows(xg[g], B * Ts, Kp, cg, 0)
            hip.gemm_nt(a, [wg[g]], [bias_g[g]] if bias_g is not None else None, out=y[g], splitk=False)
        return xg, y

    @staticmethod
    def _pack_weight(weight, G, flip):
        """[C, cg, k] -> [G, cg_out, Kp] with K index (tap, cin); flip=True gives the dgrad kernel (taps reversed, in/out swapped)."""
        C, cg, k = weight.shape
        w = weight.view(G, cg, cg, k)  # [g, co, ci, j]
        if flip:
            w = w.flip(3).permute(0, 2, 3, 1)  # [g, ci, j', co]
        else:
            w = w.permute(0, 1, 3, 2)          # [g, co, j, ci]
        w = w.reshape(G, cg, k * cg)
        Kp = (k * cg + 63) // 64 * 64
        return F.pad(w, (0, Kp - k * cg)).contiguous()

    @staticmethod
    def forward(ctx, x, weight, bias, groups):
        B, T, C = x.shape
        G, cg, k = groups, C // groups, weight.shape[2]
        assert k % 2 == 1 and cg % 8 == 0 and (k + 1) * cg >= (k * cg + 63) // 64 * 64
        x = x.contiguous()
        wg = GroupedConv1dSameFn._pack_weight(weight, G, flip=False)
        xg, y = GroupedConv1dSameFn._run(x, wg, bias.view(G, cg) if bias is not None else None, B, T, G, cg, k)
        Ts = T + k - 1
        out = y.view(G, B, Ts, cg)[:, :, :T].permute(1, 2, 0, 3).reshape(B, T, C)
        ctx.save_for_backward(xg, weight)
        ctx.dims = (B, T, C, G, cg, k, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dy):
        xg, weight = ctx.saved_tensors
        B, T, C, G, cg, k, has_bias = ctx.dims
        dy = dy.contiguous()
        Ts = T + k - 1
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wflip = GroupedConv1dSameFn._pack_weight(weight, G, flip=True)
            _, dxr = GroupedConv1dSameFn._run(dy, wflip, None, B, T, G, cg, k)
            dx = dxr.view(G, B, Ts, cg)[:, :, :T].permute(1, 2, 0, 3).reshape(B, T, C)
        if ctx.needs_input_grad[1]:
            dyr = torch.zeros(G, B * Ts, cg, dtype=dy.dtype, device=dy.device)  # output-row layout, zero garbage rows
            dyr.view(G, B, Ts, cg)[:, :, :T] = dy.view(B, T, G, cg).permute(2, 0, 1, 3)
            dwg = torch.empty(G, cg, k * cg, dtype=dy.dtype, device=dy.device)
            probs = [(dyr[g], _as_rows(xg[g], B * Ts, k * cg, cg, 0), dwg[g], False) for g in range(G)]
            # the G per-group products as ONE persistent launch (op_gemm_tn_grouped: every tile runs its whole K, no split-K slabs and
            # folds; 16 launches + 16 folds per layer before round 4) when the shapes qualify, else one launch per group
            if not (G <= hip.TN_GROUP_MAX and (B * Ts) % 64 == 0 and cg % 8 == 0 and hip.gemm_tn_grouped(probs)):
                for dyg, patches, out, _ in probs:
                    ops.wgrad(dyg, patches, out=out)
            dw = dwg.view(G, cg, k, cg).permute(0, 1, 3, 2).reshape(C, cg, k)  # [g, co, j, ci] -> [C, ci, j]
        if has_bias and ctx.needs_input_grad[2]:
            db = hip.colsum(dy.view(B * T, C))
        return dx, dw, db, None


def grouped_conv1d_same(x, weight, bias, groups):
    return GroupedConv1dSameFn.apply(x, weight, bias, groups)
