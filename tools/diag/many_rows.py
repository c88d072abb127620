import sys, time
sys.path[:0] = ['lz-ani_amd', 'oracle', 'tools']
import numpy as np, lzani_ctypes as L, oracle as O, synth_genomes as SG
st = SG.Stream(9)
n = 70000
base = (st.u64(300) % np.uint64(4)).astype(np.uint8)
seqs = []
for k in range(n):
    g = base.copy()
    idx = (st.u64(6) % np.uint64(300)).astype(np.int64)
    g[idx] = (g[idx] + 1) % 4
    seqs.append(g)
ref_ids = np.arange(n, dtype=np.uint32)
row_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(2))
q = np.stack([(np.arange(n) + 1) % n, (np.arange(n) + 7) % n], axis=1).reshape(-1).astype(np.uint32)
eng = L.Engine(); eng.set_genomes(seqs)
t = time.perf_counter(); got = eng.run_rows(ref_ids, row_off, q); dt = time.perf_counter() - t
tm = eng.timing(); eng.close()
m = 400
want = O.ref_rows(seqs, ref_ids[:m], row_off[:m + 1], q[:2 * m], None, 8)
print("rows", n, "pairs", len(q), "wall %.3f s" % dt, tm, "sample mismatches", int((got[:2 * m] != want).any(axis=1).sum()))
want2 = O.ref_rows(seqs, ref_ids[n - m:], row_off[n - m:] - row_off[n - m], q[2 * (n - m):], None, 8)
print("tail sample mismatches", int((got[2 * (n - m):] != want2).any(axis=1).sum()))
