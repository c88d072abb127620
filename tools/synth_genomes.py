"""Seeded synthetic viral/bacterial genome sets.

Self-contained counter-based PRNG (splitmix64 over an index array), vectorised with numpy, so the
byte-exact genomes depend only on (seed, n, length range) and not on numpy's Generator.
Families of `fam` genomes: member 0 is a uniform random ancestor, members 1.. are the ancestor
with per-base substitutions at rate d ~ U(dmin, dmax), indels at rate d/10 (length 1-10, half
insertions) and, with probability 0.2, one 1-5 kbp inversion (reverse complement).
Returns symbol codes (0..3), one uint8 array per genome.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, idx):
    """u64 stream element(s) idx of the splitmix64 sequence started at seed."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Stream:
    def __init__(self, seed):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.pos = 0

    def u64(self, n):
        v = splitmix64(self.seed, np.arange(self.pos, self.pos + n, dtype=np.uint64))
        self.pos += n
        return v

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def one(self):
        return float(self.uniform(1)[0])

    def randint(self, lo, hi):  # inclusive
        return lo + int(self.u64(1)[0] % np.uint64(hi - lo + 1))


def _rc(x):
    return (3 - x[::-1]).astype(np.uint8)


def mutate(anc, d, st):
    g = anc.copy()
    n = len(g)
    sub = st.uniform(n) < d
    shift = (st.u64(n) % np.uint64(3)).astype(np.uint8) + 1
    g[sub] = (g[sub] + shift[sub]) % 4
    ev = np.nonzero(st.uniform(n) < d / 10.0)[0]
    if len(ev):
        kinds = st.u64(len(ev))
        pieces, prev = [], 0
        for p, k in zip(ev.tolist(), kinds.tolist()):
            if p < prev:
                continue
            ln = 1 + (k >> 1) % 10
            pieces.append(g[prev:p])
            if k & 1:   # insertion of ln random bases
                ins = (splitmix64(k, np.arange(ln, dtype=np.uint64)) % np.uint64(4)).astype(np.uint8)
                pieces.append(ins)
                prev = p
            else:       # deletion of ln bases
                prev = min(n, p + ln)
        pieces.append(g[prev:])
        g = np.concatenate(pieces)
    if st.one() < 0.2 and len(g) > 12000:
        ln = st.randint(1000, 5000)
        s = st.randint(0, len(g) - ln)
        g[s:s + ln] = _rc(g[s:s + ln])
    return g


def make_set(n, seed, lmin=36000, lmax=44000, fam=10, dmin=0.01, dmax=0.15):
    """n genomes; names g%06d_f%d_m%d."""
    st = Stream(seed)
    seqs, names = [], []
    anc = None
    for i in range(n):
        f, m = divmod(i, fam)
        if m == 0:
            L = st.randint(lmin, lmax)
            anc = (st.u64(L) % np.uint64(4)).astype(np.uint8)
            g = anc
        else:
            d = dmin + (dmax - dmin) * st.one()
            g = mutate(anc, d, st)
        seqs.append(np.ascontiguousarray(g))
        names.append("g%06d_f%d_m%d" % (i, f, m))
    return names, seqs


def _generator_version():
    """Hash of the generator's own source: a cached set made by an older mutate/make_set is never served."""
    import hashlib, inspect
    src = "".join(inspect.getsource(f) for f in (Stream, mutate, make_set))
    return hashlib.sha256(src.encode()).hexdigest()[:12]


def make_set_cached(n, seed, cache_dir=None, **kw):
    """make_set through an on-disk cache (one .npz per argument set and generator version): the 10,000-genome bench
    set takes ~1 min to generate, and a profiling session runs the bench many times on one box.  The cache lives in a
    per-user directory (LZANI_SYNTH_CACHE, else $XDG_CACHE_HOME/lzani_synth, else ~/.cache/lzani_synth, else a
    uid-suffixed directory under /tmp); an entry carries a checksum of its codes and is regenerated when it does not
    verify."""
    import os, zipfile, zlib
    cache_dir = cache_dir or os.environ.get("LZANI_SYNTH_CACHE")
    if not cache_dir:
        base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
        cache_dir = os.path.join(base, "lzani_synth")
        try:
            os.makedirs(cache_dir, mode=0o700, exist_ok=True)
        except OSError:
            cache_dir = "/tmp/lzani_synth_cache_%d" % os.getuid()
    key = "set_n%d_s%d_%s_g%s.npz" % (n, seed, "_".join("%s%s" % (k, kw[k]) for k in sorted(kw)), _generator_version())
    path = os.path.join(cache_dir, key)
    try:
        z = np.load(path)
        off, codes = z["off"], z["codes"]
        if len(off) == n + 1 and int(z["crc"]) == zlib.crc32(codes.tobytes()):
            return [str(x) for x in z["names"]], [codes[off[i]:off[i + 1]] for i in range(n)]
    except (OSError, KeyError, ValueError, EOFError, zipfile.BadZipFile, zlib.error):
        pass                                   # no entry, a truncated / corrupt one, or not one of ours: generate (and replace it)
    names, seqs = make_set(n, seed, **kw)
    try:
        os.makedirs(cache_dir, mode=0o700, exist_ok=True)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        codes = np.concatenate(seqs) if n else np.zeros(0, np.uint8)
        tmp = path + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, off=off, codes=codes, names=np.array(names), crc=np.uint32(zlib.crc32(codes.tobytes())))
        os.replace(tmp, path)
    except OSError:
        pass                                   # (a read-only or full cache directory only costs the next run its minute)
    return names, seqs


def write_fasta(path, names, seqs, width=70):
    lut = np.frombuffer(b"ACGTNN", dtype=np.uint8)
    with open(path, "wb") as f:
        for nm, s in zip(names, seqs):
            f.write(b">" + nm.encode() + b"\n")
            txt = lut[np.minimum(s, 5)].tobytes()
            for k in range(0, len(txt), width):
                f.write(txt[k:k + width] + b"\n")


if __name__ == "__main__":
    import sys
    n, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    names, seqs = make_set(n, seed)
    write_fasta(out, names, seqs)
