#!/usr/bin/env python
"""profiles/<tag>_pmc_hbm.txt (tools/run_profile_final.sh / run_profile_r02.sh) -> profiles/<tag>_traffic.json read by bench.py.
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 128-B read requests as 64 B
(MI355X_MICROARCH.md, HBM section); both counters are in KiB per dispatch."""
import json, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_j"
txt = open(f"profiles/{tag}_pmc_hbm.txt").read()
names = {"k_conv3_r32<": "k_conv3_r32<bf16>", "k_wgrad_r32<": "k_wgrad_r32<bf16>+reduce", "k_norm_bwd_apply<": "k_norm_bwd_apply<bf16>",
         "k_norm_act_fwd<": "k_norm_act_fwd<bf16>", "k_up_tile<": "k_up_tile<bf16>", "k_conv_igemm<cbim::bf16_tag, 2, 1,": "k_conv_igemm<bf16,2,1>", "k_conv_igemm<cbim::bf16_tag, 2, 2,": "k_conv_igemm<bf16,2,2>",
         "k_conv_igemm<cbim::bf16_tag, 1, 2,": "k_conv_igemm<bf16,1,2>", "k_conv_wgrad<": "k_conv_wgrad<bf16>"}
vals = {}
for m in re.finditer(r"== (\w+) (.*?)\n\1\s+total/dispatch ([\d.e+]+)", txt):
    if m.group(2).strip() in names:
        vals.setdefault(names[m.group(2).strip()], {})[m.group(1)] = float(m.group(3))
out = {"_source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 "
                  f"--graph 0` on MI355X (tools/run_profile_final.sh / run_profile_r02.sh); raw per-dispatch averages in profiles/{tag}_pmc_hbm.txt; "
                  "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)"}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out[k] = {"fetch_kb": v["FETCH_SIZE"], "write_kb": v["WRITE_SIZE"],
                  "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024}
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
if tag.startswith("r02"):
    json.dump(out, open("profiles/r02_traffic.json", "w"), indent=1)
if tag.startswith("r03"):
    json.dump(out, open("profiles/r03_traffic.json", "w"), indent=1)      # the file bench.py reads
print(json.dumps(out, indent=1))
