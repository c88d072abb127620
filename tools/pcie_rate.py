import sys, time
sys.path[:0]=['lz-ani_amd','tools']
import numpy as np, lzani_ctypes as L, synth_genomes as SG
names,seqs=SG.make_set(1000,1)
eng=L.Engine()
t=time.perf_counter(); eng.set_genomes(seqs); t_set=time.perf_counter()-t
ref_ids,row_off=L.dense_rows(1000)
eng.run_rows(ref_ids,row_off,None)
t=time.perf_counter(); out=eng.run_rows(ref_ids,row_off,None); t_host=time.perf_counter()-t
tm=eng.timing()
print(f"set_genomes {t_set*1e3:.1f} ms; run_rows(host out) {t_host*1e3:.1f} ms wall; kernel {tm['pairs_ms']:.1f} ms + index {tm['index_ms']:.1f} ms -> {999000/t_host:.0f} pairs/s PCIe-inclusive")
