"""Manual GPU smoke script (not collected by pytest): HIP path vs oracle on small sets."""
import glob, os, sys, time, faulthandler
faulthandler.dump_traceback_later(90, exit=True)
os.environ.setdefault("LZANI_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "lz-ani_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle as O
import lzani_ctypes as L
import synth_genomes as SG

def check(name, seqs, params=None):
    print("start", name, flush=True)
    eng = L.Engine(params)
    print("engine up", flush=True)
    t = time.time(); eng.set_genomes(seqs); t_set = time.time() - t
    t = time.time(); got = eng.all2all(); t_run = time.time() - t
    tm = eng.timing()
    print("gpu done", t_set, t_run, tm, flush=True)
    t = time.time(); want = O.oracle_all2all(seqs, params, threads=16); t_or = time.time() - t
    bad = np.argwhere((got != want).any(axis=2))
    n = len(seqs)
    print(f"{name}: n={n} pairs={n*(n-1)} equal={len(bad)==0} bad={len(bad)} set={t_set:.3f}s run={t_run:.3f}s "
          f"(pairs_ms={tm['pairs_ms']:.1f} index_ms={tm['index_ms']:.1f}) oracle16={t_or:.2f}s -> {n*(n-1)/max(tm['pairs_ms'],1e-9)*1e3:.0f} pairs/s", flush=True)
    for r, q in bad[:8]:
        print("   ref", r, "qry", q, "got", got[r, q].tolist(), "want", want[r, q].tolist())
    eng.close()
    return len(bad) == 0

ok = True
ex = [s[1] for s in O.read_multifasta(os.path.join(ROOT, "tests/golden/example/multifasta.fna"))]
ok &= check("example", ex)
V = [np.concatenate([x[1] for x in O.read_multifasta(f)]) for f in sorted(glob.glob(os.path.join(ROOT, "tests/golden/vir61/*.fna")))]
ok &= check("vir61", V)
ok &= check("example mal15", ex, dict(mal=15, msl=9, reg=60))
ok &= check("example mrd20mqd60", ex, dict(mrd=20, mqd=60))
_, syn = SG.make_set(200, 1)
ok &= check("synth200", syn)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
