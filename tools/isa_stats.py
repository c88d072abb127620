"""Static instruction counts of the pair kernels in a device assembly listing (tools/isa.sh)."""
import re
import sys


def main(path, only="k_pairs"):
    txt = open(path).read()
    meta = {m.group(1): m.group(2) for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S)}
    for m in re.finditer(r"^(_ZN5lzani\w+):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if only not in name or name not in meta:
            continue
        ops = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = lambda p: sum(1 for o in ops if o.startswith(p))
        g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", meta[name]) or [None, "?"])[1]
        print(f"{name}: insts {len(ops)} valu {c('v_')} salu {c('s_')} vmem {c('global_') + c('buffer_') + c('flat_')} lds {c('ds_')} "
              f"scratch_ops {c('scratch_')} writelane {c('v_writelane')} readlane {c('v_readlane')} branches {c('s_cbranch') + c('s_branch')} "
              f"vgpr {g('next_free_vgpr')} sgpr {g('next_free_sgpr')} scratch_bytes {g('private_segment_fixed_size')}")


if __name__ == "__main__":
    main(*sys.argv[1:])
