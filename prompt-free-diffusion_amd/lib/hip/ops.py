"""Tensor-level wrappers over the C ABI (binding.py): torch is used here for device memory and
the current HIP stream only -- every arithmetic op below is a hand-written gfx950 kernel.

Layout convention of the whole product path: activations are fp16, token-major / NHWC
(`[B, H, W, C]` or `[M, C]`, channel stride 1, row stride `ld`).  NCHW exists only at the
reference API boundary (ops.to_nhwc / ops.to_nchw).
"""
import ctypes as C
import functools
import os
import threading

import torch

from . import binding as _b
from .binding import ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU  # noqa: F401

_byref = C.byref


def _lib():
    return _b.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


# The reference's models are called from Gradio worker threads on ONE shared global model with no lock
# (app.py:277, 405-410).  Everything here runs on the current stream and shares per-device scratch (the
# split-K slabs below, the per-layer packed-weight caches, captured hipGraphs with static input buffers),
# so every public entry point of the product path (composite model methods, DDIMSampler, the pipeline)
# holds this re-entrant lock for the duration of its launch sequence: concurrent requests serialise
# instead of interleaving a GEMM and its split-K reduce with another thread's GEMM.
DEVICE_LOCK = threading.RLock()


def serialised(fn):
    @functools.wraps(fn)
    def wrapper(*a, **k):
        with DEVICE_LOCK:
            return fn(*a, **k)
    return wrapper


_WS = {}
_WS_BYTES = 96 << 20


def _workspace(device):
    """per-device scratch for split-K GEMMs (the library keeps f16 partial sums in it since round 6; sized at 4 bytes per element as ABI 9 requires) (stable address for hipGraphs; launch sequences are serialised by
    DEVICE_LOCK and run on one stream, so one slab per device suffices)"""
    ws = _WS.get(device)
    if ws is None:
        ws = _WS[device] = torch.empty(_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


_TRACE = os.environ.get("PFD_TRACE_GEMM")
_TRACE_GN = os.environ.get("PFD_TRACE_GN")
# False: every LayerNorm is its own launch again (tests flip it per call; the round-3 A/B: -52 ms per batch for the fold)
LN_FOLD = True


# One record per GEMM / conv launch: THE field list of profiles/unet_c2_gemm_shapes.txt.  The writer below, the readers in
# tests/test_hip_kernels_fullsize.py (parse_launch_records) and csrc/selftest.cpp (--replay: 19, 22 or 24 integers per line) follow
# it; tests/test_host.py parses the tracked file with it on CPU, so a regenerated file cannot break a GPU-only test unseen.
TRACE_FIELDS = ("M", "N", "K", "act", "has_bias", "has_rowvec", "has_res", "bias_per_row",
                "ksize", "stride", "pad", "ups", "B", "H", "W", "Cin", "Ho", "Wo", "rows_per_rv",
                "k_split", "zero_rows", "gn_out", "gnf", "res_rows")
TRACE_FIELDS_ABI7 = 19      # rounds 1-3 wrote the first 19 (k_split = zero_rows = gn_out = 0)
TRACE_FIELDS_ABI8 = 22      # rounds 4-5 the first 22 (ABI 9's gnf = 0 | 1 keeps the raw result | 2 skips it, res_rows = 0: ADVICE r05)


def trace_record(d):
    """the TRACE_FIELDS of a PfdGemmDesc, as ints"""
    return tuple(int(v) for v in (
        d.M, d.N, d.K, d.act, d.bias is not None, d.rowvec is not None, d.R is not None, d.bias_per_row,
        d.ksize, d.stride, d.pad, d.ups, d.B, d.H, d.Wd, d.Cin, d.Ho, d.Wo, d.rows_per_rv,
        d.k_split, d.zero_rows, bool(d.gn_out), (2 if d.gnf_skip_raw else 1) if d.gnf_y else 0, d.res_rows))


def parse_launch_records(path, distinct=True):
    """[LaunchRecord] of a PFD_TRACE_GEMM file (19-field lines of older rounds are padded with zeros)"""
    import collections
    Rec = collections.namedtuple("LaunchRecord", TRACE_FIELDS)
    seen, out = set(), []
    for ln, line in enumerate(open(path), 1):
        v = tuple(int(t) for t in line.split())
        if not v:
            continue
        if len(v) in (TRACE_FIELDS_ABI7, TRACE_FIELDS_ABI8):
            v = v + (0,) * (len(TRACE_FIELDS) - len(v))
        if len(v) != len(TRACE_FIELDS):
            raise ValueError(f"{path}:{ln}: {len(v)} fields, expected {len(TRACE_FIELDS)} ({' '.join(TRACE_FIELDS)})")
        if distinct and v in seen:
            continue
        seen.add(v)
        out.append(Rec(*v))
    return out


def _trace(d):
    """PFD_TRACE_GEMM=<file>: append one line per GEMM/conv launch (shape replay for tools/selftest --replay,
    used to measure HBM traffic with rocprofv3 --pmc on a torch-free process)"""
    with open(_TRACE, "a") as f:
        f.write(" ".join(str(v) for v in trace_record(d)) + "\n")


def _chk16(t, what):
    if t.dtype != torch.float16:
        raise TypeError(f"{what}: expected float16, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP path has no CPU fallback")
    if t.stride(-1) != 1:
        raise ValueError(f"{what}: innermost stride must be 1")


def _rows(t):
    """(rows, cols, ld) of a tensor viewed as a row-major matrix with unit column stride."""
    cols = t.shape[-1]
    if t.dim() == 1:
        return 1, cols, cols
    ld = t.stride(-2)
    rows = t.numel() // cols
    # all leading dims must be collapsible onto the row stride
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise ValueError(f"tensor of shape {tuple(t.shape)} strides {t.stride()} is not a strided matrix")
        exp *= t.shape[d]
    return rows, cols, ld


# ----------------------------------------------------------------------------------------------
# GEMM / convolution
# ----------------------------------------------------------------------------------------------
def ln_fold_ok(C):
    """widths whose LayerNorm can be folded into the consumer GEMM (PfdGemmDesc.ln_stats: 160-column slices, <= 8; the
    contraction must not need K padding -- pack_matrix pads K to 64 and the fold's gamma has C entries)"""
    return LN_FOLD and C % 160 == 0 and C // 160 <= 8 and C % 64 == 0


def ln_rowstats(x, out=None):
    """partial row sums [M, C/160, 2] (fp32) of a token matrix, in the layout gemm(ln=...) takes -- for tensors that
    were not written by a gemm(ln_out=...) launch"""
    _chk16(x, "ln_rowstats x")
    M, Cc, ld = _rows(x)
    if out is None:
        out = torch.empty((M, Cc // 160, 2), dtype=torch.float32, device=x.device)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != M * (Cc // 160) * 2:
        raise ValueError("ln_rowstats: out must be a contiguous float32 [M, C/160, 2]")
    _b.check(_lib().pfd_ln_rowstats_f16(x.data_ptr(), ld, M, Cc, out.data_ptr(), _stream()), "pfd_ln_rowstats_f16")
    return out


# GroupNorm statistics from the producers (PfdGemmDesc.gn_out / pfd_groupnorm_pstats_f16); PFD_GN_PSTATS=0: every
# two-launch GroupNorm reads its input for statistics again (A/B runs)
GN_PSTATS = True


def gn_stats_wanted(B, HW, N, M=None):
    """should the launch that writes a [B, HW, N] tensor also emit GroupNorm(32) statistics of it?  Yes where a GroupNorm
    over these channels (alone or as one source of a skip concat) takes the two-launch form and the wide-tile kernels
    can form them (N = 320 | 640 | 1280, whole 64-row slabs per sample)."""
    if not GN_PSTATS or N not in (320, 640, 1280) or HW % 64 or (M is not None and M != B * HW):
        return False
    return bool(_lib().pfd_groupnorm_takes_pstats(B, N, 0, HW, 32))


def _new_gn_stats(M, N, device):
    return torch.empty((M // 64, N // 160, 16, 2), dtype=torch.float32, device=device)


def set_gn_stats(t, stats):
    """attach the producer's statistics to the tensor OBJECT that holds its output (views / copies do not inherit them)"""
    t._pfd_gn = (stats, t.data_ptr(), tuple(t.shape), t._version)
    return t


def get_gn_stats(t):
    """the statistics the producer of `t` emitted, or None (also when the object was re-pointed since)"""
    ent = getattr(t, "_pfd_gn", None)
    if ent is None or ent[1] != t.data_ptr() or ent[2] != tuple(t.shape) or ent[3] != t._version:
        return None       # (the version counter also moves on torch in-place ops and writes through a view of this tensor)
    return ent[0]


def _written(out, stats=None):
    """every op that writes into a caller-provided `out` goes through here: the tensor object carries the statistics of
    THIS write or none (stale producer statistics would be normalised with silently).  A write through a VIEW of a
    statistics-carrying tensor invalidates them through the shared version counter (get_gn_stats / get_normed compare it)."""
    if out is not None:
        # the library wrote through a raw pointer: tell torch (the version counter is shared by every view of the storage), so
        # that statistics / normalised copies riding on ANOTHER object over these bytes -- ControlNet's `add(h[b:b+1], g,
        # out=h[b:b+1])` writes through a view of h -- stop matching
        torch._C._increment_version((out,))     # (an iterable of tensors: a bare tensor is iterated row by row)
    if stats is not None:
        set_gn_stats(out, stats)
    elif out is not None and getattr(out, "_pfd_gn", None) is not None:
        out._pfd_gn = None
    if out is not None and getattr(out, "_pfd_normed", None) is not None:
        out._pfd_normed = None
    return out


def set_normed(t, key, y):
    """attach the GroupNorm of `t` that its producer already computed (conv(gn_fuse=...)): key = (id of the norm module's
    gamma storage, eps, silu).  Rides on the tensor OBJECT like the statistics; any rewrite of the object drops it."""
    t._pfd_normed = (key, t.data_ptr(), tuple(t.shape), y, t._version)
    return t


def get_normed(t, key):
    ent = getattr(t, "_pfd_normed", None)
    if ent is None or ent[0] != key or ent[1] != t.data_ptr() or ent[2] != tuple(t.shape) or ent[4] != t._version:
        return None       # (a torch in-place op or a write through a view bumps the version -> recompute)
    return ent[3]


def cat_pair(t):
    """torch.cat([t, t]) of a CFG pair, statistics included (per-sample slabs: the copy's are the original's)"""
    out = torch.cat([t, t])
    st = get_gn_stats(t)
    if st is not None:
        set_gn_stats(out, torch.cat([st, st]))
    return out


# ABI 8 forms (PfdGemmDesc.k_split / zero_rows); PFD_GEMM_FUSE=0 keeps the two-launch forms (A/B runs)
GEMM_FUSE = True


def wide_tile_ok(N, K):
    """shapes the wide-tile linear kernels take (they alone serve gemm(a2=...) / gemm(zero_rows=...))"""
    return GEMM_FUSE and (N % 160 == 0 or N % 128 == 0) and K % 64 == 0


def gemm(a, w, *, bias=None, rowvec=None, rows_per_rv=1, res=None, act=ACT_NONE, out=None,
         bias_per_row=False, n=None, k=None, tile=0, out_t=None, n_split=None, ln=None, ln_out=None,
         a2=None, zero_rows=0, gn_out=False, res_rows=None):
    """out[M, N] = epi(a[M, K] @ w[N, K]^T); see pfd_gemm_f16 in include/pfd_hip.h.
    a2: second source of the contraction -- the operand is the virtual column concat [a | a2] (K = Ka + Ka2; the
    1x1 skip convolution over a skip concat).  zero_rows: that many all-zero operand rows come in front of a's rows
    (M = zero_rows + rows of a); their result is epi(0).  Both: wide-tile kernels only (wide_tile_ok).
    gn_out: the launch also emits the GroupNorm statistics of its output (PfdGemmDesc.gn_out); they ride on the returned
    tensor (get_gn_stats).
    res_rows: `res` holds that many rows and is the residual of BOTH halves of a doubled batch (M == 2 * res_rows; the CFG
    pair [x | x] stored once, PfdGemmDesc.res_rows).  Wide-tile kernels only.
    out_t / n_split: columns >= n_split go, transposed, to out_t[N - n_split, M] (wide-tile path only).
    ln = (stats, colsum, eps): LayerNorm of `a` folded into the contraction (w is the gamma-scaled weight, bias is b';
    PfdGemmDesc.ln_stats).  ln_out: True (allocate) or a float32 [M, N/160, 2] tensor -> the partial row sums of the
    OUTPUT are written there and (out, stats) is returned."""
    _chk16(a, "gemm A")
    _chk16(w, "gemm W")
    M, Ka, lda = _rows(a)
    Nw, Kw, ldw = _rows(w)
    N = Nw if n is None else n
    K = min(Ka, Kw) if k is None else k
    if a2 is not None:
        _chk16(a2, "gemm A2")
        M2, Ka2, lda2 = _rows(a2)
        if M2 != M or a2.device != a.device:
            raise ValueError(f"gemm: a2 {tuple(a2.shape)} does not pair with a {tuple(a.shape)}")
        if k is None:
            K = Ka + Ka2
        if Ka % 64 or Ka >= K or K - Ka > Ka2:
            raise ValueError(f"gemm: a2 needs a first source of a multiple of 64 columns below K (Ka {Ka}, K {K})")
    if zero_rows:
        if zero_rows < 0 or ln is not None:
            raise ValueError("gemm: zero_rows must be >= 0 and cannot be combined with ln=")
        M += zero_rows
    n_out = N // 2 if act == ACT_GEGLU else N
    if out_t is not None:
        n_out = n_split
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=a.device)
    else:
        _chk16(out, "gemm out")
        if out.device != a.device or out.shape[-1] != n_out or out.numel() != M * n_out:
            raise ValueError(f"gemm: out {tuple(out.shape)} on {out.device} cannot hold [{M}, {n_out}] on {a.device}")
    _, _, ldc = _rows(out)
    d = _b.PfdGemmDesc()
    if out_t is not None:
        _chk16(out_t, "gemm out_t")
        if out_t.shape[0] != N - n_split or out_t.shape[1] < M or out_t.stride(1) != 1:
            raise ValueError(f"gemm: out_t {tuple(out_t.shape)} cannot hold [{N - n_split}, {M}]")
        d.Ct, d.ldct, d.n_split = out_t.data_ptr(), out_t.stride(0), n_split
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias, d.rowvec, d.R = _ptr(bias), _ptr(rowvec), _ptr(res)
    d.lda, d.ldw, d.ldc = lda, ldw, ldc
    d.ldr = _rows(res)[2] if res is not None else 0
    d.ldrv = _rows(rowvec)[2] if rowvec is not None else 0
    d.M, d.N, d.K = M, N, K
    d.rows_per_rv, d.act, d.bias_per_row = rows_per_rv, act, 1 if bias_per_row else 0
    d.ksize = 0
    d.ws, d.ws_bytes = _workspace(a.device).data_ptr(), _WS_BYTES
    if a2 is not None:
        d.A2, d.lda2, d.k_split = a2.data_ptr(), lda2, Ka
    d.zero_rows = zero_rows
    if res_rows is not None:
        if res is None or M != 2 * res_rows or _rows(res)[0] != res_rows:
            raise ValueError(f"gemm: res_rows {res_rows} needs a residual of exactly that many rows and M == 2 * res_rows (M {M})")
        d.res_rows = res_rows
    elif res is not None and _rows(res)[0] != M:
        raise ValueError(f"gemm: residual of {_rows(res)[0]} rows for {M} output rows (res_rows= names a shared one)")
    if ln is not None:
        st, cs, eps = ln
        if st.dtype != torch.float32 or not st.is_contiguous() or st.dim() != 3 or st.shape[0] != M or \
                st.shape[1] * 160 != K or st.shape[2] != 2 or st.device != a.device:
            raise ValueError(f"gemm: ln stats {tuple(st.shape)} {st.dtype} do not describe a [{M}, {K}] operand")
        if cs.dtype != torch.float32 or cs.numel() != N or not cs.is_contiguous() or cs.device != a.device:
            raise ValueError("gemm: ln column sums must be a contiguous float32 [N]")
        d.ln_stats, d.ln_colsum, d.ln_parts, d.ln_eps = st.data_ptr(), cs.data_ptr(), st.shape[1], float(eps)
    stats = None
    if ln_out is not None and ln_out is not False:
        if ln_out is True:
            stats = torch.empty((M, N // 160, 2), dtype=torch.float32, device=a.device)
        else:
            stats = ln_out
            if stats.dtype != torch.float32 or not stats.is_contiguous() or stats.numel() != M * (N // 160) * 2:
                raise ValueError("gemm: ln_out must be a contiguous float32 [M, N/160, 2]")
        d.ln_out = stats.data_ptr()
    gst = None
    if gn_out:
        gst = _new_gn_stats(M, N, a.device)
        d.gn_out = gst.data_ptr()
    if _TRACE:
        _trace(d)
    lib = _lib()
    rc = lib.pfd_gemm_f16_ex(_byref(d), tile, _stream()) if tile else lib.pfd_gemm_f16(_byref(d), _stream())
    _b.check(rc, f"pfd_gemm_f16 M{M} N{N} K{K}")
    _written(out, gst)
    return out if stats is None else (out, stats)


def conv_gn_fusable(B, H, W_, C1, C2, N, ksize=3, stride=1, pad=1):
    """True when conv(..., gn=...) is served (PfdGemmDesc.gn_table: the 3x3 patch kernel) AND pays: at 16-wide
    images GroupNorm is the single-launch small-slab kernel, cheaper than statistics + table."""
    return (ksize == 3 and stride == 1 and pad == 1 and W_ in (32, 64) and H % (256 // W_) == 0 and
            (B * H * W_) % 256 == 0 and N % 160 == 0 and C1 % 64 == 0 and C2 % 64 == 0)


# Shapes whose fused GroupNorm request (conv(gn_fuse=...)) the library declined once (it decides whether a problem splits K):
# not asked again.  Key: everything the decision depends on.
_GNF_DECLINED = set()


def conv(x, w, ksize, *, stride=1, pad=None, ups=False, bias=None, rowvec=None, res=None, act=ACT_NONE,
         out=None, tile=0, out_hw=None, rows_per_rv=None, gn=None, gn_out=False, gn_fuse=None):
    """Implicit-GEMM convolution of an NHWC image x[B,H,W,Cin] (Cin % 64 == 0) with packed
    weights w[N, ksize*ksize*Cin]; returns [B,Ho,Wo,N].  rowvec: [B, N] per-sample vector.
    gn = (table, x2, silu): GroupNorm(+SiLU) of the virtual concat [x | x2] applied while the input is staged
    (table from groupnorm_table; see PfdGemmDesc.gn_table) -- x is then the UN-normalised tensor.
    gn_fuse = (gamma, beta, eps, silu, keep_raw): GroupNorm(32)(+SiLU) of the OUTPUT inside the launch's split-K reduction
    (PfdGemmDesc.gnf_y, ABI 9).  Returns (raw | None, normalised) when the library serves it, else None with NOTHING launched
    (a problem it does not split, a width the fused reduction is not built for): the caller runs conv + groupnorm."""
    _chk16(x, "conv x")
    _chk16(w, "conv W")
    B, H, W_, Cin = x.shape
    x2 = None
    if gn is not None:
        table, x2, gn_silu = gn
        C1 = Cin
        if x2 is not None:
            _chk16(x2, "conv x2")
            if x2.shape[:3] != x.shape[:3] or x2.device != x.device:
                raise ValueError(f"conv: x2 {tuple(x2.shape)} does not match x {tuple(x.shape)}")
            ld2 = x2.stride(2)
            if x2.stride(1) != ld2 * W_ or (B > 1 and x2.stride(0) != ld2 * W_ * H):
                raise ValueError("conv: x2 must be dense over (B, H, W)")
            Cin = C1 + x2.shape[-1]
        if table.dtype != torch.float32 or table.device != x.device or tuple(table.shape) != (B, 2, Cin) or \
                not table.is_contiguous():
            raise ValueError(f"conv: gn table {tuple(table.shape)} {table.dtype} is not a float32 [{B}, 2, {Cin}]")
    if x.stride(2) < x.shape[-1]:
        raise ValueError("conv: bad pixel stride")
    lda = x.stride(2)
    if x.stride(1) != lda * W_ or (B > 1 and x.stride(0) != lda * W_ * H):
        raise ValueError("conv: image must be dense over (B, H, W)")
    if pad is None:
        pad = ksize // 2
    Hin, Win = (2 * H, 2 * W_) if ups else (H, W_)
    Ho = (Hin + 2 * pad - ksize) // stride + 1
    Wo = (Win + 2 * pad - ksize) // stride + 1
    if out_hw is not None:  # asymmetric (bottom/right) zero padding: taps past the image read 0
        Ho, Wo = out_hw
    N, K, ldw = _rows(w)
    if K != ksize * ksize * Cin:
        raise ValueError(f"conv: packed weight K {K} != {ksize}*{ksize}*{Cin}")
    M = B * Ho * Wo
    if out is None:
        out = torch.empty((B, Ho, Wo, N), dtype=torch.float16, device=x.device)
    else:
        _chk16(out, "conv out")
        if out.device != x.device or out.shape[-1] != N or out.numel() != M * N:
            raise ValueError(f"conv: out {tuple(out.shape)} on {out.device} cannot hold [{B},{Ho},{Wo},{N}]")
    d = _b.PfdGemmDesc()
    d.A, d.W, d.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias, d.rowvec, d.R = _ptr(bias), _ptr(rowvec), _ptr(res)
    d.lda, d.ldw, d.ldc = lda, ldw, _rows(out)[2]
    d.ldr = _rows(res)[2] if res is not None else 0
    d.ldrv = _rows(rowvec)[2] if rowvec is not None else 0
    d.M, d.N, d.K = M, N, K
    d.rows_per_rv, d.act, d.bias_per_row = (Ho * Wo if rows_per_rv is None else rows_per_rv), act, 0
    d.ksize, d.stride, d.pad, d.ups = ksize, stride, pad, 1 if ups else 0
    d.B, d.H, d.Wd, d.Cin, d.Ho, d.Wo = B, H, W_, Cin, Ho, Wo
    d.ws, d.ws_bytes = _workspace(x.device).data_ptr(), _WS_BYTES
    if gn is not None:
        d.gn_table, d.gn_c1, d.gn_act = table.data_ptr(), C1, ACT_SILU if gn_silu else ACT_NONE
        if x2 is not None:
            d.A2, d.lda2 = x2.data_ptr(), x2.stride(2)
    gst = None
    if gn_fuse is not None:
        if gn is not None or gn_out:
            raise ValueError("conv: gn_fuse cannot be combined with gn= / gn_out=")
        g_gamma, g_beta, g_eps, g_silu, g_keep = gn_fuse
        # (everything the library's decision can depend on -- residual present, leading dimensions, device --
        #  and only for heuristic calls: a decline under a forced tile says nothing about the heuristic's choice)
        key = (B, H, W_, Cin, N, ksize, stride, pad, bool(ups), Ho, Wo, act, rowvec is not None,
               None if rowvec is None else d.rows_per_rv, res is not None, int(d.lda), int(d.ldc), int(d.ldr), str(x.device))
        if tile == 0 and key in _GNF_DECLINED:
            return None
        _chk16(g_gamma, "conv gn_fuse gamma")
        _chk16(g_beta, "conv gn_fuse beta")
        if g_gamma.numel() != N or g_beta.numel() != N:
            raise ValueError(f"conv: gn_fuse gamma / beta must have {N} entries")
        y = torch.empty((B, Ho, Wo, N), dtype=torch.float16, device=x.device)
        d.gnf_gamma, d.gnf_beta, d.gnf_y, d.gnf_ldy = g_gamma.data_ptr(), g_beta.data_ptr(), y.data_ptr(), N
        d.gnf_eps, d.gnf_act, d.gnf_rows, d.gnf_skip_raw = float(g_eps), ACT_SILU if g_silu else ACT_NONE, Ho * Wo, 0 if g_keep else 1
        lib = _lib()
        rc = lib.pfd_gemm_f16_ex(_byref(d), tile, _stream()) if tile else lib.pfd_gemm_f16(_byref(d), _stream())
        if rc == _b.PFD_ESHAPE:      # nothing was launched (include/pfd_hip.h)
            if tile == 0:
                _GNF_DECLINED.add(key)
            return None
        _b.check(rc, f"pfd_gemm_f16(conv, fused GroupNorm) M{M} N{N} K{K}")
        if _TRACE:
            _trace(d)
        return (_written(out) if g_keep else None), y
    if gn_out and gn is None and gn_stats_wanted(B, Ho * Wo, N) and Cin % 64 == 0:   # (True = "where a GroupNorm will use them")
        gst = _new_gn_stats(M, N, x.device)
        d.gn_out = gst.data_ptr()
    if _TRACE:
        _trace(d)
    lib = _lib()
    rc = lib.pfd_gemm_f16_ex(_byref(d), tile, _stream()) if tile else lib.pfd_gemm_f16(_byref(d), _stream())
    _b.check(rc, f"pfd_gemm_f16(conv) M{M} N{N} K{K}" + (" with GroupNorm prologue" if gn is not None else ""))
    return _written(out, gst)


def im2col(x, ksize, stride, pad, kpad, ho=None, wo=None):
    """[B,H,W,Cin] -> [B*Ho*Wo, kpad] patch matrix (zero padded) for narrow-channel convs."""
    _chk16(x, "im2col x")
    B, H, W_, Cin = x.shape
    Ho = (H + 2 * pad - ksize) // stride + 1 if ho is None else ho
    Wo = (W_ + 2 * pad - ksize) // stride + 1 if wo is None else wo
    col = torch.empty((B * Ho * Wo, kpad), dtype=torch.float16, device=x.device)
    rc = _lib().pfd_im2col_f16(x.data_ptr(), x.stride(2), col.data_ptr(), B, H, W_, Cin, ksize, stride, pad,
                               Ho, Wo, kpad, _stream())
    _b.check(rc, "pfd_im2col_f16")
    return col, Ho, Wo


def conv_narrow(x, w, ksize, *, stride=1, pad=None, bias=None, rowvec=None, res=None, act=ACT_NONE,
                ho=None, wo=None, out=None, gn_out=False):
    """Convolution whose Cin is not a multiple of 64: im2col + GEMM.  w: [N, kpad] packed."""
    if pad is None:
        pad = ksize // 2
    B = x.shape[0]
    col, Ho, Wo = im2col(x, ksize, stride, pad, w.shape[1], ho, wo)
    N = w.shape[0]
    o2 = None if out is None else out.view(-1, out.shape[-1])
    gn_out = bool(gn_out) and gn_stats_wanted(B, Ho * Wo, N) and wide_tile_ok(N, w.shape[1])
    y = gemm(col, w, bias=bias, rowvec=rowvec, rows_per_rv=Ho * Wo, res=res, act=act, out=o2, gn_out=gn_out)
    r = y.view(B, Ho, Wo, N) if out is None else out
    return _written(r, get_gn_stats(y) if gn_out else None)


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
def attention(q, k, vt, B, H, Nq, Nk, D, scale, *, ldq, ldk, ldvt, q_bs, k_bs, vt_bs, out=None):
    """Fused attention; see pfd_attention_f16.  q/k/vt may be views into wider buffers."""
    if out is None:
        out = torch.empty((B * Nq, H * D), dtype=torch.float16, device=q.device)
    d = _b.PfdAttnDesc()
    d.Q, d.K, d.Vt, d.O = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.ldq, d.ldk, d.ldvt, d.ldo = ldq, ldk, ldvt, out.stride(-2)
    d.q_bs, d.k_bs, d.vt_bs, d.o_bs = q_bs, k_bs, vt_bs, Nq * out.stride(-2)
    d.B, d.H, d.Nq, d.Nk, d.D = B, H, Nq, Nk, D
    d.scale = scale
    _b.check(_lib().pfd_attention_f16(_byref(d), _stream()), f"pfd_attention_f16 B{B} H{H} Nq{Nq} Nk{Nk} D{D}")
    return out


def swin_window_attention(qkv, qkv_bias, rpb, B, H, W_, Cdim, nH, ws, shift, scale):
    _chk16(qkv, "swin qkv")
    out = torch.empty((B * H * W_, Cdim), dtype=torch.float16, device=qkv.device)
    d = _b.PfdSwinAttnDesc()
    d.qkv, d.qkv_bias, d.rpb, d.out = qkv.data_ptr(), qkv_bias.data_ptr(), rpb.data_ptr(), out.data_ptr()
    d.B, d.H, d.W, d.C, d.nH, d.ws, d.shift = B, H, W_, Cdim, nH, ws, shift
    d.scale = scale
    _b.check(_lib().pfd_swin_window_attention_f16(_byref(d), _stream()), "pfd_swin_window_attention_f16")
    return out


# ----------------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------------
def groupnorm(x, gamma, beta, groups, eps, *, x2=None, silu=False, out=None):
    """GroupNorm(+SiLU) of NHWC x[B,H,W,C1] (optionally virtually concatenated with x2[B,H,W,C2]).  When the launches
    that wrote x (and x2) emitted their statistics (gemm / conv gn_out=True) and the shape qualifies, the normalisation is
    ONE launch from those sums (pfd_groupnorm_pstats_f16); otherwise statistics + apply (or the small-slab kernel)."""
    _chk16(x, "groupnorm x")
    B = x.shape[0]
    C1 = x.shape[-1]
    HW = x.numel() // (B * C1)
    C2 = 0 if x2 is None else x2.shape[-1]
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (C1 + C2,), dtype=torch.float16, device=x.device)
    lib = _lib()
    st1 = get_gn_stats(x) if GN_PSTATS else None
    st2 = get_gn_stats(x2) if (GN_PSTATS and x2 is not None) else None
    if _TRACE_GN:   # PFD_TRACE_GN=<file>: which norms find their producers' statistics (tools/gn_paths.py)
        with open(_TRACE_GN, "a") as f:
            f.write(f"{B} {HW} {C1} {C2} {int(st1 is not None)} {int(x2 is None or st2 is not None)} "
                    f"{int(bool(lib.pfd_groupnorm_takes_pstats(B, C1, C2, HW, groups)))}\n")
    if st1 is not None and (x2 is None or st2 is not None) and groups == 32 and \
            lib.pfd_groupnorm_takes_pstats(B, C1, C2, HW, groups):
        rc = lib.pfd_groupnorm_pstats_f16(x.data_ptr(), C1, x.stride(-2), st1.data_ptr(), _ptr(x2), C2,
                                          0 if x2 is None else x2.stride(-2), _ptr(st2), gamma.data_ptr(), beta.data_ptr(),
                                          out.data_ptr(), out.stride(-2), B, HW, groups, eps,
                                          ACT_SILU if silu else ACT_NONE, _stream())
        _b.check(rc, f"pfd_groupnorm_pstats_f16 B{B} HW{HW} C{C1}+{C2}")
        return _written(out)
    wsb = lib.pfd_groupnorm_ws_bytes(B, C1 + C2, HW)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    rc = lib.pfd_groupnorm_f16(x.data_ptr(), C1, x.stride(-2), _ptr(x2), C2, 0 if x2 is None else x2.stride(-2),
                               gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), out.stride(-2), B, HW, groups,
                               eps, ACT_SILU if silu else ACT_NONE, ws.data_ptr(), wsb, _stream())
    _b.check(rc, f"pfd_groupnorm_f16 B{B} HW{HW} C{C1}+{C2}")
    return _written(out)


def groupnorm_table(x, gamma, beta, groups, eps, *, x2=None):
    """GroupNorm statistics of NHWC x[B,H,W,C1] (| x2) as the per-(sample, channel) affine map the convolution's
    GroupNorm prologue applies: float32 [B, 2, C1+C2] = (rstd * gamma | beta - mean * rstd * gamma)."""
    _chk16(x, "groupnorm_table x")
    B = x.shape[0]
    C1 = x.shape[-1]
    HW = x.numel() // (B * C1)
    C2 = 0 if x2 is None else x2.shape[-1]
    if x2 is not None:
        _chk16(x2, "groupnorm_table x2")
    table = torch.empty((B, 2, C1 + C2), dtype=torch.float32, device=x.device)
    lib = _lib()
    wsb = lib.pfd_groupnorm_ws_bytes(B, C1 + C2, HW)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    rc = lib.pfd_groupnorm_table_f16(x.data_ptr(), C1, x.stride(-2), _ptr(x2), C2, 0 if x2 is None else x2.stride(-2),
                                     gamma.data_ptr(), beta.data_ptr(), table.data_ptr(), B, HW, groups, eps,
                                     ws.data_ptr(), wsb, _stream())
    _b.check(rc, f"pfd_groupnorm_table_f16 B{B} HW{HW} C{C1}+{C2}")
    return table


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _chk16(x, "layernorm x")
    M, Cdim, ldx = _rows(x)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    rc = _lib().pfd_layernorm_f16(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                  _rows(out)[2], M, Cdim, eps, 0, 0, 0, 0, _stream())
    _b.check(rc, f"pfd_layernorm_f16 M{M} C{Cdim}")
    return _written(out)


def layernorm_patch_merge(x, gamma, beta, eps=1e-5):
    """PatchMerging gather + LayerNorm(4C): x[B,H,W,C] -> [B*ceil(H/2)*ceil(W/2), 4C]."""
    _chk16(x, "layernorm_patch_merge x")
    B, H, W_, Cq = x.shape
    Ho, Wo = (H + 1) // 2, (W_ + 1) // 2
    out = torch.empty((B * Ho * Wo, 4 * Cq), dtype=torch.float16, device=x.device)
    rc = _lib().pfd_layernorm_f16(x.data_ptr(), x.stride(2), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                  4 * Cq, B * Ho * Wo, 4 * Cq, eps, 1, B, H, W_, _stream())
    _b.check(rc, "pfd_layernorm_f16(gather4)")
    return out


def softmax_rows(x, scale, out=None):
    _chk16(x, "softmax x")
    R, N, ldx = _rows(x)
    if out is None:
        out = torch.empty((R, N), dtype=torch.float16, device=x.device)
    _b.check(_lib().pfd_softmax_rows_f16(x.data_ptr(), ldx, out.data_ptr(), _rows(out)[2], R, N, scale, _stream()),
             "pfd_softmax_rows_f16")
    return out


# ----------------------------------------------------------------------------------------------
# boundary / elementwise
# ----------------------------------------------------------------------------------------------
def to_nhwc(x, mul=1.0, add=0.0, rep=1):
    """NCHW fp32|fp16 -> NHWC fp16 (x*mul+add), batch repeated `rep` times."""
    if not x.is_cuda:
        raise RuntimeError("to_nhwc: input must live on the GPU; the HIP path has no CPU fallback")
    if x.dtype not in (torch.float32, torch.float16):
        x = x.float()
    x = x.contiguous()
    B, Cc, H, W_ = x.shape
    out = torch.empty((rep * B, H, W_, Cc), dtype=torch.float16, device=x.device)
    rc = _lib().pfd_nchw_to_nhwc_f16(x.data_ptr(), 1 if x.dtype == torch.float32 else 0, out.data_ptr(), B, Cc, H,
                                     W_, mul, add, rep, _stream())
    _b.check(rc, "pfd_nchw_to_nhwc_f16")
    return out


def to_nchw(x, dtype=torch.float16, mul=1.0, add=0.0, lo=-65504.0, hi=65504.0):
    """NHWC fp16 (dense) -> NCHW fp32|fp16, y = clamp(x*mul+add, lo, hi)."""
    _chk16(x, "to_nchw x")
    B, H, W_, Cc = x.shape
    if not x.is_contiguous():
        raise ValueError("to_nchw: dense NHWC expected")
    out = torch.empty((B, Cc, H, W_), dtype=torch.float32 if dtype == torch.float32 else torch.float16,
                      device=x.device)
    rc = _lib().pfd_nhwc_to_nchw(x.data_ptr(), out.data_ptr(), 1 if dtype == torch.float32 else 0, B, Cc, H, W_,
                                 mul, add, lo, hi, _stream())
    _b.check(rc, "pfd_nhwc_to_nchw")
    return out if out.dtype == dtype else out.to(dtype)


def image_u8(x, mul=1.0, add=0.0, f16_image=True):
    """NHWC fp16 [B,H,W,C] -> uint8 [B,H,W,C]: uint8(clamp(x*mul+add, 0, 1) * 255), the reference's ToPILImage
    arithmetic (app.py:273-275) on the device; f16_image: the image the reference would hold is fp16 (both the
    image and the product are rounded to f16 before the truncation), else fp32"""
    _chk16(x, "image_u8 x")
    if not x.is_contiguous():
        raise ValueError("image_u8: dense NHWC expected")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _b.check(_lib().pfd_image_u8_f16(x.data_ptr(), out.data_ptr(), x.numel(), float(mul), float(add),
                                     1 if f16_image else 0, _stream()),
             "pfd_image_u8_f16")
    return out


def timestep_embedding(t, dim, max_period=10000.0):
    t = t.to(torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=torch.float16, device=t.device)
    _b.check(_lib().pfd_timestep_embedding_f16(t.data_ptr(), out.data_ptr(), t.shape[0], dim, max_period, _stream()),
             "pfd_timestep_embedding_f16")
    return out


def cfg_ddim_step(eps, nb, x, coef, *, noise=None, want_next=True, rep=None):
    """Fused CFG combine + DDIM update.  eps NHWC f16 [nb*B,h,w,C]; x NCHW fp32 [B,C,h,w];
    coef fp32[5] device.  Returns (x_prev fp32 NCHW, pred_x0 fp32 NCHW, xin_next f16 NHWC|None);
    xin_next holds `rep` copies of the batch (default nb: the CFG-doubled UNet input)."""
    B, Cc, h, w = x.shape
    rep = nb if rep is None else rep
    x_prev = torch.empty_like(x)
    pred_x0 = torch.empty_like(x)
    xin = torch.empty((rep * B, h, w, Cc), dtype=torch.float16, device=x.device) if want_next else None
    rc = _lib().pfd_cfg_ddim_step(eps.data_ptr(), nb, x.data_ptr(), _ptr(noise), coef.data_ptr(), x_prev.data_ptr(),
                                  pred_x0.data_ptr(), _ptr(xin), rep, B, Cc, h, w, _stream())
    _b.check(rc, "pfd_cfg_ddim_step")
    return x_prev, pred_x0, xin


def add(a, b, out=None):
    _chk16(a, "add a")
    _chk16(b, "add b")
    if not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("add: dense tensors expected")
    if a.numel() != b.numel() or (out is not None and out.numel() != a.numel()):
        raise ValueError(f"add: operand sizes differ ({tuple(a.shape)} vs {tuple(b.shape)}); no broadcasting")
    if out is None:
        out = torch.empty_like(a)
    _b.check(_lib().pfd_add_f16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "pfd_add_f16")
    return _written(out)


def axpby(a, alpha, b=None, beta=0.0, out=None):
    """out = alpha*a + beta*b (b optional)"""
    _chk16(a, "axpby a")
    if not a.is_contiguous() or (b is not None and not b.is_contiguous()):
        raise ValueError("axpby: dense tensors expected")
    if (b is not None and b.numel() != a.numel()) or (out is not None and out.numel() != a.numel()):
        raise ValueError("axpby: operand sizes differ; no broadcasting")
    if out is None:
        out = torch.empty_like(a)
    _b.check(_lib().pfd_axpby_f16(a.data_ptr(), float(alpha), _ptr(b), float(beta), out.data_ptr(), a.numel(),
                                  _stream()), "pfd_axpby_f16")
    return _written(out)


def add_rowvec(x, v, out=None, ln_out=None):
    """out = x + v[None, :]; ln_out: a float32 [R, C/160, 2] tensor that receives the partial row sums of `out` in the
    same launch (the layout gemm(ln=...) takes)"""
    _chk16(x, "add_rowvec x")
    R, Cc, ldx = _rows(x)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    if ln_out is not None:
        if ln_out.dtype != torch.float32 or not ln_out.is_contiguous() or ln_out.numel() != R * (Cc // 160) * 2:
            raise ValueError("add_rowvec: ln_out must be a contiguous float32 [R, C/160, 2]")
        _b.check(_lib().pfd_add_rowvec_lnstats_f16(x.data_ptr(), ldx, v.data_ptr(), out.data_ptr(), _rows(out)[2], R, Cc,
                                                   ln_out.data_ptr(), _stream()), "pfd_add_rowvec_lnstats_f16")
        return _written(out)
    _b.check(_lib().pfd_add_rowvec_f16(x.data_ptr(), ldx, v.data_ptr(), out.data_ptr(), _rows(out)[2], R, Cc,
                                       _stream()), "pfd_add_rowvec_f16")
    return _written(out)


def activation(x, act, out=None):
    _chk16(x, "activation x")
    if not x.is_contiguous():
        raise ValueError("activation: dense tensor expected")
    if out is None:
        out = torch.empty_like(x)
    _b.check(_lib().pfd_act_f16(x.data_ptr(), out.data_ptr(), x.numel(), act, _stream()), "pfd_act_f16")
    return _written(out)
