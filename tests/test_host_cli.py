"""The C++ host binary `lz-ani` (lz-ani_amd/host): FASTA ingest, reorder, kmer-db filter and TSV emit
(SURVEY 8(f) rows N1/N2/N4).  CPU tests feed it the matching-stage integers through its --results-in
test seam (oracle numbers; the binary itself has no CPU compute path); the GPU tests run it end to end."""
import os
import subprocess

import numpy as np
import pytest

import oracle as O
import synth_genomes as SG
import util as U

EXE = os.path.join(U.ROOT, "lz-ani_amd", "host", "lz-ani")


@pytest.fixture(scope="module", autouse=True)
def build_host():
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])


def _raw_file(path, res, pairs=None):
    n = res.shape[0]
    with open(path, "w") as f:
        for r in range(n):
            for q in range(n):
                if r != q and (pairs is None or (r, q) in pairs):
                    f.write(f"{r} {q} {res[r, q, 0]} {res[r, q, 1]} {res[r, q, 2]}\n")


def _oracle_reordered(loader, params=None):
    names, seqs = U.reorder(*loader())
    return names, seqs, O.oracle_all2all(seqs, params, threads=8)


def run(args, **kw):
    return subprocess.run([EXE] + args, capture_output=True, text=True, **kw)


def test_vir61_golden_files_byte_exact(tmp_path):
    """BASELINE config 1: --in-dir test/vir61, default format -> test/vir61.ani.tsv + .ids.tsv."""
    names, seqs, res = _oracle_reordered(U.load_vir61)
    raw = str(tmp_path / "raw.txt")
    _raw_file(raw, res)
    out = str(tmp_path / "ani.tsv")
    p = run(["all2all", "--in-dir", os.path.join(U.GOLD, "vir61"), "--out", out, "--results-in", raw])
    assert p.returncode == 0, p.stderr
    assert open(out).read() == open(os.path.join(U.GOLD, "vir61.ani.tsv")).read()
    assert open(str(tmp_path / "ani.ids.tsv")).read() == open(os.path.join(U.GOLD, "vir61.ani.ids.tsv")).read()


def test_example_golden_files_and_formats(tmp_path):
    names, seqs, res = _oracle_reordered(U.load_example)
    raw = str(tmp_path / "raw.txt")
    _raw_file(raw, res)
    fa = os.path.join(U.GOLD, "example", "multifasta.fna")
    out = str(tmp_path / "ani.tsv")
    assert run(["all2all", "--in-fasta", fa, "-o", out, "--results-in", raw, "-V", "0"]).returncode == 0
    assert open(out).read() == open(os.path.join(U.GOLD, "example", "ani.tsv")).read()
    assert open(str(tmp_path / "ani.ids.tsv")).read() == open(os.path.join(U.GOLD, "example", "ani.ids.tsv")).read()
    lens = [len(s) for s in seqs]
    complete = "qidx,ridx,query,reference,tani,gani,ani,qcov,rcov,num_alns,len_ratio,qlen,rlen,nt_match,nt_mismatch".split(",")
    for fmt, cols, pct in (("complete", complete, False), ("lite", "qidx,ridx,tani,gani,ani,qcov,num_alns,len_ratio".split(","), False),
                           ("nt_match,query,rcov", ["nt_match", "query", "rcov"], True)):
        o2 = str(tmp_path / "x.out")
        args = ["all2all", "--in-fasta", fa, "-o", o2, "--results-in", raw, "--out-format", fmt, "--out-ids", str(tmp_path / "my.ids")]
        if pct:
            args += ["--out-in-percent", "true"]
        assert run(args).returncode == 0
        assert open(o2).read() == U.emit_tsv(names, lens, res, cols, in_percent=pct)
        assert os.path.exists(str(tmp_path / "my.ids"))


def test_out_filter_and_single_txt(tmp_path):
    names, seqs, res = _oracle_reordered(U.load_example)
    raw = str(tmp_path / "raw.txt")
    _raw_file(raw, res)
    fa = os.path.join(U.GOLD, "example", "multifasta.fna")
    out = str(tmp_path / "f.tsv")
    assert run(["all2all", "--in-fasta", fa, "-o", out, "--results-in", raw, "--out-filter", "ani", "0.9", "--out-filter", "qcov", "0.5",
                "--out-format", "complete"]).returncode == 0
    rows = [ln.split("\t") for ln in open(out).read().split("\n")[1:] if ln]
    full = [ln.split("\t") for ln in U.emit_tsv(names, [len(s) for s in seqs], res,
            "qidx,ridx,query,reference,tani,gani,ani,qcov,rcov,num_alns,len_ratio,qlen,rlen,nt_match,nt_mismatch".split(",")).split("\n")[1:] if ln]
    keep = [r for r in full if int(r[13]) / max(1, int(r[13]) + int(r[14])) >= 0.9 and (int(r[13]) + int(r[14])) / int(r[11]) >= 0.5]
    assert rows == keep and 0 < len(rows) < len(full)
    one = str(tmp_path / "one.txt")
    assert run(["all2all", "--in-fasta", fa, "-o", one, "--results-in", raw, "--out-type", "single-txt"]).returncode == 0
    txt = open(one).read()
    assert txt.startswith("[params]\nmin_anchor_len             : 11\n") and "[lz_similarities]\n" in txt
    body = txt.split("[lz_similarities]\n")[1].strip().split("\n")
    assert len(body) == 66
    a, b, m1, l1, c1, m2, l2, c2 = (int(x) for x in body[0].split())
    assert (a, b) == (0, 1) and [m1, l1, c1] == res[1, 0].tolist() and [m2, l2, c2] == res[0, 1].tolist()


def test_kmerdb_filter_rows(tmp_path):
    """--flt-kmerdb example/fltr.txt 0.9 keeps 13 unordered pairs = 26 TSV rows."""
    names, seqs, res = _oracle_reordered(U.load_example)
    raw = str(tmp_path / "raw.txt")
    _raw_file(raw, res)
    fa = os.path.join(U.GOLD, "example", "multifasta.fna")
    out = str(tmp_path / "flt.tsv")
    p = run(["all2all", "--in-fasta", fa, "-o", out, "--results-in", raw, "--flt-kmerdb", os.path.join(U.GOLD, "example", "fltr.txt"), "0.9"])
    assert p.returncode == 0 and "Filter size: 26" in p.stderr
    # with --results-in every pair is present, so the emit is the full table; the filter count is what is checked here
    bad = run(["all2all", "--in-fasta", os.path.join(U.GOLD, "vir61", os.listdir(os.path.join(U.GOLD, "vir61"))[0]), "-o", out,
               "--results-in", raw, "--flt-kmerdb", os.path.join(U.GOLD, "example", "fltr.txt"), "0.9"])
    assert bad.returncode == 1 and "different size" in bad.stderr


def test_cli_conventions(tmp_path):
    assert run(["--version"]).stderr.strip() == "1.2.3"
    p = run([])
    assert p.returncode == 0 and "Usage:" in p.stderr               # usage + return 0 (lz-ani.cpp:341-342)
    assert run(["all2all", "--bogus", "1"]).returncode == 1         # unknown parameter: exit(1)
    assert run(["frobnicate", "x", "y"]).returncode == 0            # unknown mode: usage, return 0
    p = run(["all2all", "--in-fasta", "/nonexistent.fna", "-o", str(tmp_path / "o.tsv")])
    assert p.returncode == 1 and "Cannot open file" in p.stderr


def test_parameter_envelope_is_refused_up_front(tmp_path):
    """The reference accepts any ints for the LZ knobs (lz-ani.cpp:205-260); this engine's wave formulation has an
    envelope (64-bit lane masks), and the binary says which flag is outside it before it reads any input."""
    for flag, val in (("--mqd", "100"), ("--aw", "65"), ("--ar", "70"), ("--mal", "40"), ("--msl", "0"), ("--am", "-1")):
        p = run(["all2all", "--in-fasta", str(tmp_path / "does_not_exist.fna"), "-o", str(tmp_path / "o.tsv"), flag, val])
        assert p.returncode == 1 and f"Unsupported value: {flag} {val}" in p.stderr, (flag, p.stderr)
    p = run(["all2all", "--in-fasta", str(tmp_path / "does_not_exist.fna"), "-o", str(tmp_path / "o.tsv"), "--mqd", "64", "--aw", "64"])
    assert "Unsupported" not in p.stderr and "Cannot open file" in p.stderr


def test_ingest_quirks(tmp_path):
    """Name cut at the first space, lower case accepted, CRLF, last unterminated line dropped in
    multi-FASTA mode, contigs joined by mrd N's when --multisample-fasta false (seq_len includes them)."""
    fa = tmp_path / "q.fna"
    fa.write_bytes(b">s1 some description\r\nACGTACGTAC\r\nacgtnn\r\n>s2\nACGT\nACGTAC")
    raw = tmp_path / "raw.txt"
    raw.write_text("0 1 0 0 0\n1 0 0 0 0\n")
    out = tmp_path / "o.tsv"
    assert run(["all2all", "--in-fasta", str(fa), "-o", str(out), "--results-in", str(raw)]).returncode == 0
    assert (tmp_path / "o.ids.tsv").read_text() == "id\tseq_len\tno_parts\ns1\t16\t1\ns2\t4\t1\n"
    raw.write_text("")                                            # one item only: no pairs
    assert run(["all2all", "--in-fasta", str(fa), "-o", str(out), "--results-in", str(raw), "--multisample-fasta", "false"]).returncode == 0
    assert (tmp_path / "o.ids.tsv").read_text() == "id\tseq_len\tno_parts\nq.fna\t66\t1\n"       # 16 + 40 + 10
    # the streaming parser at its seams: bases before the first header and after a bare '>' are dropped, empty lines
    # and a lone CR line are ignored, a CR inside a line is a symbol (N), an unterminated header line opens nothing
    fa.write_bytes(b"ACGT\n>a x\nAC\n\n\r\nGT\n>\nTTTT\n>b\nA\rC\nGG\n>c")
    raw.write_text("0 1 0 0 0\n1 0 0 0 0\n")
    assert run(["all2all", "--in-fasta", str(fa), "-o", str(out), "--results-in", str(raw)]).returncode == 0
    assert (tmp_path / "o.ids.tsv").read_text() == "id\tseq_len\tno_parts\nb\t5\t1\na\t4\t1\n"
    # a file larger than the 4 MB read window, CRLF line ends straddling it
    big = tmp_path / "big.fna"
    big.write_bytes(b">x\r\n" + b"ACGTACGTAC\r\n" * 500_000 + b">y\r\nAC\r\n")
    assert run(["all2all", "--in-fasta", str(big), "-o", str(out), "--results-in", str(raw)]).returncode == 0
    assert (tmp_path / "o.ids.tsv").read_text() == "id\tseq_len\tno_parts\nx\t5000000\t1\ny\t2\t1\n"


def test_number_formatting_against_reference_and_restatement():
    """The emitter's real_to_chars (lz-ani_amd/host/emit.h) against the Python restatement used by the golden
    rebuilds and, where oracle/_ref exists, against the reference's own refresh::real_to_pchar."""
    import ctypes as C
    import random
    import struct
    d = os.path.join(U.ROOT, "tests", "model")
    so = os.path.join(d, "libemit_shim.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", os.path.join(d, "emit_shim.cpp"), "-o", so, "-lz"])
    lib = C.CDLL(so)
    lib.host_format_real.argtypes = [C.c_double, C.c_int, C.c_char_p]
    buf = C.create_string_buffer(64)
    have_ref = O.lib_ref() is not None and hasattr(O.lib_ref(), "ref_format_real")
    rnd = random.Random(3)
    vals = [0.0, 1.0, 0.5, 1e-7, 4e-7, 0.000364, 0.98725, 0.999999, 0.9999995, 0.99999949, 9.9999995, 123456.5,
            1234567.0, 1e21, 1e22, 5e-324, 1.7976931348623157e308]
    for _ in range(20000):
        k = rnd.random()
        if k < 0.5:
            v = rnd.randint(0, 10 ** rnd.randint(1, 7)) / rnd.randint(1, 10 ** rnd.randint(1, 7))
        elif k < 0.7:
            v = 100.0 * rnd.randint(0, 50000) / rnd.randint(1, 50000)
        elif k < 0.9:
            v = rnd.random() * 10 ** rnd.randint(-12, 12)
        else:
            v = struct.unpack("d", struct.pack("Q", rnd.getrandbits(62)))[0]
        if v == v and v != float("inf"):
            vals.append(v)
    for i, v in enumerate(vals):
        prec = (4, 6, 6, 6, 1, 2, 9, 15)[i % 8]
        n = lib.host_format_real(v, prec, buf)
        mine = buf.raw[:n].decode()
        assert mine == U.real_to_str(v, prec), (v, prec)
        if have_ref:
            assert mine == O.ref_format_real(v, prec), (v, prec)


@pytest.mark.skipif(O.lib_ref() is None or not hasattr(O.lib_ref(), "ref_expand_output_format"), reason="oracle/_ref not built")
def test_out_format_expansion_matches_reference(tmp_path):
    """--out-format meta names and column lists: the header the binary writes is what CParams derives."""
    names, seqs, res = _oracle_reordered(U.load_example)
    raw = str(tmp_path / "raw.txt")
    _raw_file(raw, res)
    fa = os.path.join(U.GOLD, "example", "multifasta.fna")
    for fmt in ("standard", "lite", "complete", "lite,rlen,qlen", "nt_match,standard", "query,reference,tani", "rcov"):
        out = str(tmp_path / "f.tsv")
        assert run(["all2all", "--in-fasta", fa, "-o", out, "--results-in", raw, "--out-format", fmt]).returncode == 0
        assert open(out).readline().rstrip("\n").split("\t") == O.ref_expand_output_format(fmt).split(",")
    assert O.ref_expand_output_format("tani,bogus") == "!bogus"
    p = run(["all2all", "--in-fasta", fa, "-o", str(tmp_path / "g.tsv"), "--results-in", raw, "--out-format", "tani,bogus"])
    assert p.returncode == 0 and "Unknown output-format component: bogus" in p.stderr      # parse failure: return 0, as the reference


def test_gz_and_in_txt_inputs(tmp_path):
    """gzip-compressed FASTA (file_wrapper.h:472-606 in the reference) and --in-txt give the same files."""
    import gzip
    names, seqs, res = _oracle_reordered(U.load_example)
    raw = str(tmp_path / "raw.txt")
    _raw_file(raw, res)
    fa = os.path.join(U.GOLD, "example", "multifasta.fna")
    gz = str(tmp_path / "multi.fna.gz")
    with open(fa, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    lst = tmp_path / "list.txt"
    lst.write_text(gz + "\n")
    for args in (["--in-fasta", gz], ["--in-txt", str(lst)]):
        out = str(tmp_path / "o.tsv")
        assert run(["all2all"] + args + ["-o", out, "--results-in", raw]).returncode == 0
        assert open(out).read() == open(os.path.join(U.GOLD, "example", "ani.tsv")).read()


@pytest.mark.gpu
def test_end_to_end_on_gpu(tmp_path):
    """The whole binary on the GPU: vir61 (config 1) and the example set with and without the filter."""
    out = str(tmp_path / "ani.tsv")
    p = run(["all2all", "--in-dir", os.path.join(U.GOLD, "vir61"), "--out", out, "-V", "2"])
    assert p.returncode == 0, p.stderr
    assert open(out).read() == open(os.path.join(U.GOLD, "vir61.ani.tsv")).read()
    assert open(str(tmp_path / "ani.ids.tsv")).read() == open(os.path.join(U.GOLD, "vir61.ani.ids.tsv")).read()
    fa = os.path.join(U.GOLD, "example", "multifasta.fna")
    p = run(["all2all", "--in-fasta", fa, "-o", out])
    assert p.returncode == 0 and open(out).read() == open(os.path.join(U.GOLD, "example", "ani.tsv")).read()
    p = run(["all2all", "--in-fasta", fa, "-o", out, "--flt-kmerdb", os.path.join(U.GOLD, "example", "fltr.txt"), "0.9", "--out-format", "complete"])
    assert p.returncode == 0, p.stderr
    rows = [ln for ln in open(out).read().split("\n")[1:] if ln]
    assert len(rows) == 26
    full = set(U.emit_tsv(*(lambda n, s: (n, [len(x) for x in s], O.oracle_all2all(s, None, threads=8)))(*U.reorder(*U.load_example())),
                          "qidx,ridx,query,reference,tani,gani,ani,qcov,rcov,num_alns,len_ratio,qlen,rlen,nt_match,nt_mismatch".split(",")).split("\n"))
    assert all(r in full for r in rows)
    p = run(["all2all", "--in-fasta", fa, "-o", out, "--mal", "15", "--msl", "9", "--reg", "60"])
    names, seqs = U.reorder(*U.load_example())
    want = U.emit_tsv(names, [len(s) for s in seqs], O.oracle_all2all(seqs, dict(mal=15, msl=9, reg=60), threads=8), U.STANDARD)
    assert p.returncode == 0 and open(out).read() == want
    assert run(["all2all", "--in-fasta", fa, "-o", out, "--mqd", "100"]).returncode == 1      # outside the envelope: clean failure
    # --out-alignment: example/output/ani.aln.tsv as a multiset of rows (the reference's row order is thread-dependent)
    aln = str(tmp_path / "ani.aln.tsv")
    p = run(["all2all", "--in-fasta", fa, "-o", out, "--out-alignment", aln])
    assert p.returncode == 0, p.stderr
    got = open(aln).read().split("\n")
    gold = open(os.path.join(U.GOLD, "example", "ani.aln.tsv")).read().split("\n")
    assert got[0] == gold[0] and sorted(got[1:]) == sorted(gold[1:])
    assert open(out).read() == open(os.path.join(U.GOLD, "example", "ani.tsv")).read()


@pytest.mark.gpu
def test_tiled_matching_with_overlapped_emit(tmp_path):
    """A large dense all2all goes to the engine in row x column blocks and the rows of a finished block are written
    while the next block is matched (do_matching_tiled): forced here at vir61's size with blocks of 16 rows -- the same
    bytes as the golden file (= the untiled run), for the standard TSV, the complete format with an --out-filter, and
    single-txt; one more set (150 genomes, blocks of 40) against the untiled run of the same binary."""
    env = dict(os.environ, LZANI_TILE_MIN="1", LZANI_TILE_ROWS="16")
    out = str(tmp_path / "ani.tsv")
    p = run(["all2all", "--in-dir", os.path.join(U.GOLD, "vir61"), "--out", out, "-V", "2"], env=env)
    assert p.returncode == 0, p.stderr
    assert "blocks of 16 rows" in p.stderr
    assert open(out).read() == open(os.path.join(U.GOLD, "vir61.ani.tsv")).read()
    assert open(str(tmp_path / "ani.ids.tsv")).read() == open(os.path.join(U.GOLD, "vir61.ani.ids.tsv")).read()
    names, seqs = SG.make_set(150, 31, lmin=17000, lmax=20000, fam=10)
    fa = str(tmp_path / "set.fna")
    with open(fa, "w") as f:
        for nm, sq in zip(names, seqs):
            f.write(">" + nm + "\n" + "".join("ACGT"[c] for c in sq) + "\n")
    for extra in ([], ["--out-format", "complete", "--out-filter", "ani", "0.2"], ["--out-type", "single-txt"]):
        os.makedirs(str(tmp_path / "a"), exist_ok=True)
        os.makedirs(str(tmp_path / "b"), exist_ok=True)
        p = run(["all2all", "--in-fasta", fa, "-o", "out.txt"] + extra, env=dict(os.environ, LZANI_TILE_ROWS="0"), cwd=str(tmp_path / "a"))
        assert p.returncode == 0, p.stderr
        p = run(["all2all", "--in-fasta", fa, "-o", "out.txt", "-V", "2"] + extra, env=dict(os.environ, LZANI_TILE_MIN="1", LZANI_TILE_ROWS="40"),
                cwd=str(tmp_path / "b"))                  # (single-txt echoes the file name among the parameters: the same one in both runs)
        assert p.returncode == 0 and "blocks of 40 rows" in p.stderr, p.stderr
        a, b = str(tmp_path / "a" / "out.txt"), str(tmp_path / "b" / "out.txt")
        assert open(a).read() == open(b).read(), extra
        assert len(open(a).read()) > 1000


def _synth5(tmp_path, n, seed, lmin, lmax, maxfam):
    """tools/synth5.cpp: heavy-tailed families + kmer-db filter file + binary sidecar of the codes."""
    import json
    exe = str(tmp_path / "synth5")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(U.ROOT, "tools", "synth5.cpp")])
    fa, flt, side = str(tmp_path / "c5.fna"), str(tmp_path / "c5.flt"), str(tmp_path / "c5.bin")
    info = json.loads(subprocess.check_output([exe, str(n), str(seed), str(lmin), str(lmax), fa, flt, side, str(maxfam), "0.3"]))
    hdr = np.fromfile(side, dtype=np.uint64, count=n + 2)
    assert int(hdr[0]) == n
    off = hdr[1:].astype(np.int64)
    codes = np.memmap(side, dtype=np.uint8, mode="r", offset=8 * (n + 2))
    return fa, flt, info, lambda i: np.array(codes[off[i]:off[i + 1]])


def test_synth5_filter_file_format(tmp_path):
    """CPU: the config-5 generator writes what CFilter::load_filter reads (filter.cpp:34-42, 61-81): header with
    the names, rows `name,idx:val,...,` with 1-based indices of earlier genomes; the host binary's own filter
    reader reports 2 x kept pairs."""
    fa, flt, info, seq = _synth5(tmp_path, 300, 9, 800, 1200, 60)
    lines = open(flt).read().split("\n")
    names = lines[0].split(",")
    assert names[0].startswith("kmer-length:") and names[-1] == "" and len(names) == 302
    kept = 0
    for i, ln in enumerate(lines[1:301]):
        parts = ln.split(",")
        assert parts[0] == names[1 + i] and parts[-1] == ""
        for it in parts[1:-1]:
            idx, val = it.split(":")
            assert 1 <= int(idx) <= i
            kept += float(val) >= 0.3
    assert kept == info["pairs_kept"] and info["genomes"] == 300
    # through the host binary's reader (matching stage skipped: empty results-in)
    raw = tmp_path / "raw.txt"
    raw.write_text("")
    p = run(["all2all", "--in-fasta", fa, "-o", str(tmp_path / "o.tsv"), "--flt-kmerdb", flt, "0.3", "--results-in", str(raw)])
    assert p.returncode == 0 and f"Filter size: {2 * kept}" in p.stderr, p.stderr[-400:]


@pytest.mark.gpu
def test_config5_heavy_tailed_filter_20k_genomes(tmp_path):
    """BASELINE configs[4] shape through the whole binary: 20,000 genomes of ~40 kbp in heavy-tailed families
    (40 % singletons ... 5 % of 101-1000 members) with a synthetic kmer-db file, --flt-kmerdb at 0.3: the filter
    size is twice the kept pairs, every kept pair appears in both directions, and sampled rows equal the oracle."""
    n = 20_000
    fa, flt, info, seq = _synth5(tmp_path, n, 4, 36000, 44000, 1000)
    out = str(tmp_path / "ani.tsv")
    p = run(["all2all", "--in-fasta", fa, "-o", out, "--flt-kmerdb", flt, "0.3", "-V", "2",
             "--out-format", "query,reference,nt_match,nt_mismatch,num_alns"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert f"Filter size: {2 * info['pairs_kept']}" in p.stderr
    assert info["largest_row"] > 500
    import collections
    rows = collections.Counter()
    sample = []
    with open(out) as f:
        assert f.readline().rstrip("\n") == "query\treference\tnt_match\tnt_mismatch\tnum_alns"
        for k, ln in enumerate(f):
            rows["lines"] += 1
            if (k * 2654435761) % 2**32 < 2**32 // 20000:                 # ~1 line in 20,000
                sample.append(ln.rstrip("\n").split("\t"))
    assert rows["lines"] == 2 * info["pairs_kept"]
    assert len(sample) > 200
    for qn, rn, mat, lit, aln in sample[:400]:
        qi, ri = int(qn[1:7]), int(rn[1:7])
        assert O.oracle_pair(seq(ri), seq(qi)) == (int(mat), int(lit), int(aln)), (qn, rn)


@pytest.mark.gpu
def test_multi_gpu_group_through_the_binary(tmp_path, monkeypatch):
    """`lz-ani --gpus n` = lzani_group_run_rows: rows partitioned (cyclic / LPT), a context and a host thread per device,
    shards gathered on the first device, one copy out.  Rehearsed on this one-GPU box with LZANI_DEVICE_LIST=0,0,0 (three
    shards on GPU 0; the shards then move by device copies, RCCL refuses duplicate devices): the TSV must be byte-identical
    to the single-context run, dense and filtered."""
    fa = os.path.join(U.GOLD, "example", "multifasta.fna")
    flt = os.path.join(U.GOLD, "example", "fltr.txt")
    outs = {}
    for tag, env in (("one", None), ("three", "0,0,0")):
        if env:
            monkeypatch.setenv("LZANI_DEVICE_LIST", env)
        for kind, extra in (("dense", []), ("flt", ["--flt-kmerdb", flt, "0.9", "--out-format", "complete"])):
            out = str(tmp_path / f"{tag}_{kind}.tsv")
            p = run(["all2all", "--in-fasta", fa, "-o", out, "-V", "2"] + extra)
            assert p.returncode == 0, p.stderr[-800:]
            outs[(tag, kind)] = open(out).read()
            if env:
                assert p.stderr.count("GPU 0:") == 3, p.stderr[-600:]
    assert outs[("one", "dense")] == outs[("three", "dense")] == open(os.path.join(U.GOLD, "example", "ani.tsv")).read()
    assert outs[("one", "flt")] == outs[("three", "flt")] and outs[("one", "flt")].count("\n") == 27
