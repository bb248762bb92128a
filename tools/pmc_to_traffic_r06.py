#!/usr/bin/env python
"""gpurun_out/<tag>_pmc_<model>_{FETCH_SIZE,WRITE_SIZE}.json (tools/pmc_dump.py, written by tools/run_profile_r06.sh) ->
profiles/r06_traffic[_<model>].json, the files bench.py reads for roofline.traffic / roofline.step_traffic_ratio.
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM
section); both counters are KiB per dispatch.  Rows: bench.py's roofline rows under the kernels' real names (k_conv3_rw = its full
launches, the split-K launches of the 16^3 / 8^3 levels are a row of their own) + `_step`: EVERY kernel of one eager step summed.
    python tools/pmc_to_traffic_r06.py <tag> <steps traced (warm-up + timed)> [dir]"""
import json, os, re, sys
tag, steps = sys.argv[1], int(sys.argv[2])
src = sys.argv[3] if len(sys.argv) > 3 else "profiles"
FAM = [("k_conv3_rw48<bf16>", lambda k: k.startswith("k_conv3_rw48<")), ("k_map_gemm", lambda k: k.startswith("k_map_gemm")),
       ("k_resnorm_*", lambda k: k.startswith("k_resnorm")),
       ("k_conv3_rw<bf16>", lambda k: k.startswith("k_conv3_rw<") and k.rstrip().endswith("false>")),
       ("k_conv3_rw_splitk<bf16>", lambda k: k.startswith("k_conv3_rw<") and k.rstrip().endswith("true>")),
       ("k_wgrad_r32<bf16>+reduce", lambda k: k.startswith("k_wgrad_r32")),
       ("k_norm_bwd_apply<bf16>", lambda k: k.startswith("k_norm_bwd_apply<")), ("k_norm_act_fwd<bf16>", lambda k: k.startswith("k_norm_act_fwd<")),
       ("k_up_tile<bf16>", lambda k: k.startswith("k_up_tile<")), ("k_splitk_finish<bf16>", lambda k: k.startswith("k_splitk_finish<")),
       ("k_conv_igemm<bf16,2,2>", lambda k: k.startswith("k_conv_igemm<bf16_tag, 2, 2,")), ("k_conv_igemm<bf16,1,2>", lambda k: k.startswith("k_conv_igemm<bf16_tag, 1, 2,")),
       ("k_conv_wgrad<bf16>+reduce", lambda k: k.startswith("k_conv_wgrad<") or k.startswith("k_wgrad_reduce")), ("k_conv_pw<bf16>", lambda k: k.startswith("k_conv_pw<")),
       ("k_pw_wgrad<bf16>", lambda k: k.startswith("k_pw_wgrad<")), ("k_dwconv3<bf16>", lambda k: k.startswith("k_dwconv3")), ("k_winattn_*", lambda k: k.startswith("k_winattn")),
       ("k_stats_finalize", lambda k: k.startswith("k_stats_finalize")), ("k_adamw_ema", lambda k: k.startswith("k_adamw_ema")),
       ("k_lin_adjoint_axis<bf16>", lambda k: k.startswith("k_lin_adjoint_axis<")), ("hipBLASLt Cijk_*", lambda k: k.startswith("Cijk_")),
       ("ATen at::native::*", lambda k: k.startswith("at::native"))]
for model, suffix in (("resunet", ""), ("medformer", "_medformer"), ("swin_unetr", "_swin_unetr")):
    try:
        f = json.load(open(os.path.join(src, f"{tag}_pmc_{model}_FETCH_SIZE.json")))
        w = json.load(open(os.path.join(src, f"{tag}_pmc_{model}_WRITE_SIZE.json")))
    except FileNotFoundError:
        continue
    out = {"_source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --model {model} --steps {steps - 1} --warmup 1 "
                      f"--graph 0 --no-roofline --no-cpu-baseline --secondary 0` on MI355X ({steps} eager steps traced); per-kernel totals in profiles/{tag}_pmc_{model}_*.json; "
                      "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)"}
    tot_f = sum(v["sum"] for v in f.values()); tot_w = sum(v["sum"] for v in w.values())
    n_disp = sum(v["dispatches"] for v in f.values())
    out["_step"] = {"hbm_bytes_per_step": (2 * tot_f + tot_w) * 1024 / steps, "fetch_kb_per_step": tot_f / steps, "write_kb_per_step": tot_w / steps,
                    "dispatches_per_step": n_disp / steps, "steps_traced": steps}
    for name, pred in FAM:
        ks = [k for k in f if pred(k)]
        if not ks:
            continue
        nf = sum(f[k]["dispatches"] for k in ks)
        sf = sum(f[k]["sum"] for k in ks); sw = sum(w[k]["sum"] for k in ks if k in w)
        # "+reduce" rows: per launch of the main kernel (the reduce launches' bytes belong to the same gradient)
        main = [k for k in ks if not ("reduce" in k and name.endswith("+reduce"))]
        nl = sum(f[k]["dispatches"] for k in main) or nf
        out[name] = {"fetch_kb": sf / nl, "write_kb": sw / nl, "dispatches_in_trace": nl, "hbm_bytes_per_launch": (2 * sf + sw) * 1024 / nl,
                     "hbm_bytes_per_step": (2 * sf + sw) * 1024 / steps}
    if "k_conv3_rw<bf16>" in out:
        out["k_conv3_r32<bf16>"] = dict(out["k_conv3_rw<bf16>"], alias_of="k_conv3_rw<bf16>")   # the name rounds 2-4 recorded this row under
    json.dump(out, open(f"profiles/r06_traffic{suffix}.json", "w"), indent=1)
    print(model, "step GB", round(out["_step"]["hbm_bytes_per_step"] / 1e9, 2), {k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items() if not k.startswith("_") and "alias_of" not in v})
