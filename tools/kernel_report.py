#!/usr/bin/env python3
"""Summarise the AMDGPU metadata of a -save-temps .s file: VGPR/SGPR/scratch/LDS per kernel,
plus mnemonic counts that matter for the exact-order kernels (v_fma/v_mac must be absent there)."""
import re, sys, collections
src = open(sys.argv[1]).read()
kern = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", src)
meta = {}
for blk in src.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    meta[g("name")] = dict(agpr=blk.split("\n")[0].strip(), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                           spill=g("vgpr_spill_count"), lds=g("group_segment_fixed_size"))
counts = {}
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", src, re.M | re.S):
    counts[m.group(1)] = collections.Counter(re.findall(r"^\s+([a-z][a-z_0-9]+)", m.group(2), re.M))
for name, m in meta.items():
    c = counts.get(name, {})
    fm = sum(v for k, v in c.items() if k.startswith(("v_fma", "v_mac", "v_fmac", "v_pk_fma", "v_mad_f")))
    mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    print(f"{name[:70]:70s} vgpr={m['vgpr']:>4} agpr={m['agpr']:>3} sgpr={m['sgpr']:>3} scratch={m['scratch']:>4} spill={m['spill']:>3} lds={m['lds']:>6} fma={fm:>4} mfma={mf:>4} "
          f"ds_r128={c.get('ds_read_b128',0)} gl_x4={c.get('global_load_dwordx4',0)} scratch_ops={sum(v for k,v in c.items() if k.startswith('scratch_'))}")
