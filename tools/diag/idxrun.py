import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "lz-ani_amd"), os.path.join(ROOT, "tools")]
import numpy as np, lzani_ctypes as L, synth_genomes as SG
_, seqs = SG.make_set(1000, 1)
eng = L.Engine(); eng.set_genomes(seqs)
ref_ids = np.arange(1000, dtype=np.uint32); row_off = np.arange(1001, dtype=np.uint64); q = ((np.arange(1000) + 1) % 1000).astype(np.uint32)
for _ in range(3):
    try: eng.run_rows(ref_ids, row_off, q)
    except Exception as e: print(e)
