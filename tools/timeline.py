"""Where does a tile launch spend its time?  Needs the timeline build of the library:
    make -C rustyhgi_amd/csrc VARIANT=_tl EXTRA=-DHGI_TIMELINE
    HGI_LIB_PATH=rustyhgi_amd/libhgi_hip_tl.so python tools/timeline.py [c4|c3|WxHxFxL ...]
Every interior block of k_enc_tiles / k_dec_tiles logs (100 MHz s_memrealtime): first instruction, loads issued, staging
loads landed, last store acknowledged, and the XCC / CU it ran on.  Printed per launch: the span, how the starts are spread (dispatch ramp),
block lifetimes early / middle / late, blocks in flight over time, the rate at which tiles retire in the head, the
steady part and the tail, and when each XCD ran dry.  (The extra s_waitcnt vmcnt(0) in front of the two later stamps
costs a few per cent; this build is for the shape of the launch, not for its absolute time.)"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
L.hgi_debug_timeline.argtypes = [ctypes.c_void_p]
L.hgi_debug_timeline.restype = None
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
TICK_US = 0.01


def analyse(name, tl, tile_bytes, ev_us):
    tl = tl[tl[:, 0] != 0]
    if not len(tl):
        print("%s: no interior blocks logged" % name); return
    t0, t1, t2, te = (tl[:, i].astype(np.int64) for i in (0, 1, 2, 4))
    xcc = (tl[:, 3] >> 32) & 0xF
    hw = tl[:, 3] & 0xFFFFFFFF
    base = te.min()
    en, s, st, e = ((t - base) * TICK_US for t in (te, t0, t1, t2))
    span = e.max()
    n = len(tl)
    print("\n== %s: %d interior blocks, span %.1f us (first wave on a CU -> last store acknowledged); hipEvent time of the whole call %.1f us" % (name, n, span, ev_us))
    first_end = e.min()
    resident0 = int((en < first_end).sum())
    es = np.sort(en)
    print("   dispatch: %d blocks are on a CU before the first block ends (%.2f us) = %.1f waves per CU; the first 1000 / 2000 / 4000 arrive by %.2f / %.2f / %.2f us; last arrival %.1f us" % (
        resident0, first_end, resident0 / 256.0, es[min(999, n - 1)], es[min(1999, n - 1)], es[min(3999, n - 1)], en.max()))
    order = np.argsort(en)
    for label, sel in (("first round", order[:resident0]), ("middle third", order[n // 3: 2 * n // 3]), ("last round", order[-resident0:])):
        life, pro, lat = (e - en)[sel], (s - en)[sel], (st - s)[sel]
        print("   %-12s lifetime p10/p50/p90 %.1f / %.1f / %.1f us; prologue (arguments, table, tile index) p50 %.2f p90 %.2f us; "
              "staging loads landed p50 %.1f (p90 %.1f) us after their issue" % (
                  label, np.percentile(life, 10), np.median(life), np.percentile(life, 90), np.median(pro), np.percentile(pro, 90),
                  np.median(lat), np.percentile(lat, 90)))
    # how long a slot stays empty: per (XCC, HW_ID) the waves run one after another in each slot; approximate by matching,
    # per CU, every arrival with the latest earlier end on that CU that has not been matched yet
    cu = (xcc << 32) | (hw & 0x00000F00) | ((hw >> 13) & 0x7) << 16 | ((hw >> 16) & 0xF) << 20      # CU_ID, SH_ID, SE_ID
    gaps = []
    for c in np.unique(cu)[:64]:
        m = cu == c
        arr, end = np.sort(en[m]), np.sort(e[m])
        # i-th arrival beyond the first round reuses the slot freed by the (i - r)-th end, r = resident waves of this CU
        r = int((arr < end[0]).sum())
        if len(arr) > r:
            gaps.append(arr[r:] - end[:len(arr) - r])
    if gaps:
        g = np.concatenate(gaps)
        print("   slot refill (arrival of the i-th wave of a CU - end of its (i - resident)-th): p10/p50/p90 %.2f / %.2f / %.2f us (64 CUs sampled)" % (
            np.percentile(g, 10), np.median(g), np.percentile(g, 90)))
    edges = np.arange(0, np.ceil(span) + 1)
    started = np.searchsorted(np.sort(en), edges, side="right")
    ended = np.searchsorted(np.sort(e), edges, side="right")
    inflight = started - ended
    rate = np.diff(ended) * tile_bytes / 1e6      # TB/s of algorithmic bytes retired in each microsecond
    steady = np.median(rate[len(rate) // 4: 3 * len(rate) // 4])
    head = int(np.argmax(rate >= 0.8 * steady))
    tail0 = int(np.floor(en.max()))
    mid = rate[head:tail0]
    print("   steady retire rate %.2f TB/s (median of the middle half); whole span %.2f TB/s" % (steady, n * tile_bytes / span / 1e6))
    print("   head: %d us until tiles retire at >= 80 %% of that (%.1f %% of the tiles retired there); tail: %.1f us after the last "
          "block arrived (%.1f %% of the tiles retire there, mean %.2f TB/s)" % (
              head, 100.0 * ended[head] / n, span - tail0, 100.0 * (n - ended[tail0]) / n,
              (n - ended[tail0]) * tile_bytes / max(span - tail0, 1e-9) / 1e6))
    if len(mid):
        print("   between them: %.1f us at mean %.2f TB/s" % (tail0 - head, mid.mean()))
    ideal = n * tile_bytes / steady / 1e6
    print("   the same tiles at the steady rate throughout: %.1f us -> head + tail cost %.1f us" % (ideal, span - ideal))
    lasts = [e[xcc == x].max() for x in range(8) if (xcc == x).any()]
    cnt = [int((xcc == x).sum()) for x in range(8)]
    print("   XCDs: blocks %s; last store at %s us" % (cnt, " ".join("%.1f" % v for v in lasts)))
    step = max(1, len(edges) // 24)
    print("   in flight @us: " + " ".join("%d:%d" % (edges[i], inflight[i]) for i in range(0, len(edges), step)))
    print("   TB/s     @us: " + " ".join("%d:%.1f" % (edges[i], rate[i]) for i in range(0, len(rate), step)))
    fine = min(len(edges), 25)
    print("   first us, in flight: " + " ".join("%d:%d" % (edges[i], inflight[i]) for i in range(fine)))


def run(W, Hh, F, levels, quant):
    n = W * Hh
    lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
    _ffi.check(L.hgi_linear_lut(quant, lut.ctypes.data, err.ctypes.data))
    planes = H.Planes(ctx, F * n, 3)
    img, grid, out = (planes.torch(i, (F, Hh, W)) for i in range(3))
    _ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 4, 0, W, Hh, img.data_ptr(), F, n))
    nblocks = F * ((W + 127) // 128) * ((Hh + 31) // 32) + 64
    tl = torch.zeros((nblocks, 8), dtype=torch.int64, device="cuda")
    def enc(): _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, levels, 1, lut.ctypes.data, grid.data_ptr(), F, n))
    def dec(): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, levels, 1, out.data_ptr(), F, n))
    L.hgi_debug_timeline(None)
    for _ in range(40): enc(); dec()
    torch.cuda.synchronize()
    for name, fn in (("encode", enc), ("decode", dec)):
        # the alternating pattern of the bench step; the logged launch is the last of its kind
        for _ in range(6): enc(); dec()
        if name == "decode": enc()
        tl.zero_()
        torch.cuda.synchronize()
        L.hgi_debug_timeline(ctypes.c_void_p(tl.data_ptr()))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        L.hgi_debug_timeline(None)
        host = tl.cpu().numpy().astype(np.uint64)
        # (the small launches of a deep pyramid's lattice plane log into the same rows and are overwritten by the main launch)
        rows = host
        analyse("%s %d x %dx%d L%d" % (name, F, W, Hh, levels), rows, 2 * 128 * 64, a.elapsed_time(b) * 1e3)
    planes.close()


for spec in (sys.argv[1:] or ["c4", "c3"]):
    if spec == "c4": run(16384, 16384, 1, 8, 3)
    elif spec == "c3": run(4096, 4096, 64, 4, 2)
    else:
        W, Hh, F, lv = (int(v) for v in spec.split("x")); run(W, Hh, F, lv, 2)
