#!/usr/bin/env python3
"""Developer tool: per-workgroup time line of the likelihood kernel from a TM_TRACE build
(TAMCMC_ACCEL_LIB=gpurun_variants/lib_trace.so TAMCMC_TRACE_FILE=/tmp/t.bin python tools/block_trace.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tamcmc_amd
from tamcmc_amd import synth

tf = os.environ["TAMCMC_TRACE_FILE"]
w = synth.workload_c2()
n = 64
P = synth.chain_params(w, n); T = synth.temperatures(n)
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
if len(sys.argv) > 1 and sys.argv[1] == "tiny":
    q = 7 + 2 + 21 + 6 + 7 + 10
    P[:, q + 1] = 0.05
acc = tamcmc_amd.Accel(2, w["plength"], w["x"], y)
grad = os.environ.get("TAMCMC_TRACE_GRAD") is not None
if grad:
    acc.set_vars(w["index_to_relax"])
for _ in range(5):
    acc.eval_batch(P, T, grad=grad)
raw = np.fromfile(tf, dtype=np.uint64)
# launch grid: rows = blockIdx.y, columns = blockIdx.x.  Default (tile-major) order: row = rank of the tile in the
# chain's costliest-first order, column = chain; TAMCMC_ORDER=0: row = chain, column = tile slot.
rows, cols = int(raw[0]), int(raw[1])
t = raw[2:].reshape(rows, cols, 4)
if os.environ.get("TAMCMC_ORDER") != "0":
    t = t.transpose(1, 0, 2)        # -> [chain][rank]
chains, tiles = t.shape[0], t.shape[1]
t0 = t[..., 0].min()
start = (t[..., 0] - t0).astype(np.float64) * 0.01      # wall_clock64: 100 MHz -> us
mid = (t[..., 1] - t0).astype(np.float64) * 0.01
end = (t[..., 2] - t0).astype(np.float64) * 0.01
hw = t[..., 3]
xcc = (hw >> np.uint64(32)).astype(int)
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(int)
dur = end - start
print(f"tiles {tiles} chains {chains}: kernel span {end.max():.1f} us; block duration mean {dur.mean():.2f} median {np.median(dur):.2f} "
      f"p10 {np.percentile(dur,10):.2f} p90 {np.percentile(dur,90):.2f} max {dur.max():.2f}; epilogue (stamp1->2) mean {(end-mid).mean():.2f}")
print("first block start", start.min(), "last block start", start.max(), "; starts at t<1us:", (start < 1).sum(), " t<5us:", (start < 5).sum())
# concurrency over time
ts = np.linspace(0, end.max(), 25)
conc = [(np.sum((start <= x) & (end > x))) for x in ts]
print("concurrency:", " ".join(f"{int(x)}" for x in conc))
print("blocks per XCC:", np.bincount(xcc.ravel(), minlength=8))
order = np.argsort(start.ravel())
flat_chain, flat_tile = np.unravel_index(order, start.shape)
print("dispatch order (first 12 by start): ", [(int(c), int(tl)) for c, tl in zip(flat_chain[:12], flat_tile[:12])])
per_tile = dur.mean(axis=0)
print("mean duration per tile index / rank:", np.round(per_tile, 1))
print("mean start per tile index / rank:", np.round(start.mean(axis=0), 1))
print("mean end per tile index / rank:", np.round(end.mean(axis=0), 1))
print("pass 1 share of a block (stamp0->1):", round(float(((mid - start) / dur).mean()), 3))
