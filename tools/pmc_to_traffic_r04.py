#!/usr/bin/env python
"""profiles/<tag>_pmc_hbm_<model>.txt (tools/run_profile_r04.sh / tools/r04/run_pmc.sh) -> profiles/r04_traffic[_<model>].json, the
files bench.py reads for roofline.traffic.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 128-B read
requests as 64 B (MI355X_MICROARCH.md, HBM section); both counters are KiB per dispatch, averaged over the dispatches whose kernel
name contains the pattern.  The conv family is split the way bench.py's roofline rows are: the 47 full launches of k_conv3_rw
(template argument SK = false) and the split-K launches of the 16^3 / 8^3 levels (SK = true)."""
import json, re, sys
tag = sys.argv[1]
names = {"false>(cbim::R32Params)": "k_conv3_r32<bf16>", "true>(cbim::R32Params)": "k_conv3_rw_splitk<bf16>", "k_wgrad_r32<": "k_wgrad_r32<bf16>+reduce",
         "k_norm_bwd_apply<": "k_norm_bwd_apply<bf16>", "k_norm_act_fwd<": "k_norm_act_fwd<bf16>", "k_up_tile<": "k_up_tile<bf16>",
         "k_splitk_finish<": "k_splitk_finish<bf16>", "k_conv_igemm<cbim::bf16_tag, 2, 2,": "k_conv_igemm<bf16,2,2>",
         "k_conv_igemm<cbim::bf16_tag, 1, 2,": "k_conv_igemm<bf16,1,2>", "k_conv_wgrad<": "k_conv_wgrad<bf16>", "k_dwconv3": "k_dwconv3<bf16>",
         "k_winattn": "k_winattn_*"}
for model, suffix in (("resunet", ""), ("medformer", "_medformer"), ("swin_unetr", "_swin_unetr")):
    try:
        txt = open(f"profiles/{tag}_pmc_hbm_{model}.txt").read()
    except FileNotFoundError:
        continue
    vals = {}
    for m in re.finditer(r"== (\w+) (.*?)\n\1\s+total/dispatch ([\d.e+]+)\s+dispatches (\d+)", txt):
        if m.group(2).strip() in names:
            vals.setdefault(names[m.group(2).strip()], {})[m.group(1)] = (float(m.group(3)), int(m.group(4)))
    out = {"_source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --model {model} --steps 2 --warmup 1 "
                      f"--graph 0` on MI355X; raw per-dispatch averages in profiles/{tag}_pmc_hbm_{model}.txt; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) "
                      "* 1024 (gfx950 FETCH_SIZE correction); k_conv3_r32<bf16> = the full (non split-K) k_conv3_rw / k_conv3_r32 launches"}
    for k, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[k] = {"fetch_kb": v["FETCH_SIZE"][0], "write_kb": v["WRITE_SIZE"][0], "dispatches_in_trace": v["FETCH_SIZE"][1],
                      "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024}
    json.dump(out, open(f"profiles/r04_traffic{suffix}.json", "w"), indent=1)
    print(model, {k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items() if k != "_source"})
