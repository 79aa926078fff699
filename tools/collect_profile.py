#!/usr/bin/env python3
"""Profiles: from a capture on the GPU box to the tracked evidence, in two steps that cannot drift apart.

  tools/collect_profile.py fold gpurun_out/prof_<tag> <tag>
      copies what tools/profile_run.sh <tag> collected (rocprofv3 kernel-trace stats, the FETCH_SIZE / WRITE_SIZE and SQ
      counter passes, every bench line) into profiles/<tag>_* and refreshes profiles/traffic.json (the HBM-side bytes per
      step launch that bench.py reports as roofline.traffic when the kernel sources are the ones that were measured).
  tools/collect_profile.py report <tag>
      reads ONLY the tracked files profiles/<tag>_* (+ traffic.json) and writes profiles/<tag>_summary.md — every figure in
      that file is computed here, so text and CSV agree by construction (tests/test_profiles_report.py regenerates it and
      compares).  Hand-written context lives in profiles/<round>_notes.md, never in the generated file.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
ALGO = 19 * (1 << 20)                 # algorithmic bytes per step launch at 2^20 lanes
PEAK = 8e12                           # MI355X HBM3E bytes/s (/opt/skills/guides/MI355X_MICROARCH.md)
N = 1 << 20

KT = {"kt": "kernel_stats", "kt_full": "kernel_stats_full", "kt_slip": "kernel_stats_slip0p2", "kt_k20": "kernel_stats_driver_shape",
      "kt_venv": "kernel_stats_venv", "kt_other": "kernel_stats_other"}
LINES = {"bench.json": "bench", "bench_full.json": "bench_full", "bench_unprofiled.json": "bench_unprofiled",
         "bench_driver_shape.json": "bench_driver_shape", "bench_slip_unprofiled.json": "bench_slip_unprofiled",
         "bench_2rank_rehearsal.json": "bench_2rank_rehearsal", "bench_driver_shape_torch_runtime.json": "bench_driver_shape_torch_runtime",
         "bench_driver_shape_again.json": "bench_driver_shape_again", "bench_k20_profiled.json": "bench_driver_shape_profiled",
         "venv_profiled.json": "venv_profiled", "others.json": "others"}


def csrc_sha():
    h = hashlib.sha256(); d = os.path.join(ROOT, "gym_soccer_littman94_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", "Makefile")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def _find(src, sub, pattern):
    return sorted(glob.glob(os.path.join(src, sub, "runc", pattern)) + glob.glob(os.path.join(src, sub, pattern)))


def _kname(s):
    s = re.sub(r"^void ", "", s)
    return s[:s.index("(")] if "(" in s else s


def fold(src, tag):
    for sub, name in KT.items():
        files = _find(src, sub, "*kernel_stats.csv")
        if files:
            shutil.copy(files[0], os.path.join(PROF, "%s_%s.csv" % (tag, name)))
    for fn, name in LINES.items():
        p = os.path.join(src, fn)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if lines:
                open(os.path.join(PROF, "%s_%s.json" % (tag, name)), "w").write(lines[-1] + "\n")
    out, means = [["run", "kernel", "counter", "dispatches", "mean_KB", "min_KB", "max_KB"]], {}
    for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = _find(src, kind, "*counter_collection.csv")
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if "soccer::" in r["Kernel_Name"]:
                agg[_kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            out.append(["bench", k, ctr, len(v), "%.1f" % (sum(v) / len(v)), "%.1f" % min(v), "%.1f" % max(v)])
            means[(k, ctr)] = sum(v) / len(v)
    if len(out) > 1:
        csv.writer(open(os.path.join(PROF, "%s_pmc_summary.csv" % tag), "w")).writerows(out)
        tf = os.path.join(PROF, "traffic.json")
        t = json.load(open(tf))
        k = [x for x in means if "step_kernel_swar<0" in x[0]][0][0]
        t.update(round=int(tag[1:3]) if tag[:1] == "r" and tag[1:3].isdigit() else t.get("round"), build=tag, csrc_sha256=csrc_sha(),
                 commit=subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip(),
                 algorithmic_bytes_per_launch=ALGO, kernel=k, FETCH_SIZE_KB_mean=means[(k, "FETCH_SIZE")], WRITE_SIZE_KB_mean=means[(k, "WRITE_SIZE")],
                 step_kernel_hbm_bytes_per_launch=(2 * means[(k, "FETCH_SIZE")] + means[(k, "WRITE_SIZE")]) * 1024)
        json.dump(t, open(tf, "w"), indent=1)
        json.dump(t, open(os.path.join(PROF, "%s_traffic.json" % tag), "w"), indent=1)      # the tag's own copy: what `report <tag>` reads
    rows, waves = [["kernel", "counter", "dispatches", "mean_per_dispatch", "per_wave"]], {}
    aggs = []
    for sq in ("sq2", "sq1", "sq_slip"):
        files = _find(src, sq, "*counter_collection.csv")
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if "soccer::" in r["Kernel_Name"]:
                agg[(_kname(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            if c == "SQ_WAVES":
                waves[k] = sum(v) / len(v)
        aggs.append(agg)
    for agg in aggs:
        for (k, c), v in sorted(agg.items()):
            m = sum(v) / len(v)
            rows.append([k, c, len(v), "%.0f" % m, "%.1f" % (m / waves[k]) if waves.get(k) else ""])
    if len(rows) > 1:
        csv.writer(open(os.path.join(PROF, "%s_sq_counters.csv" % tag), "w")).writerows(rows)
    print("folded %s into profiles/%s_*; now: tools/collect_profile.py report %s" % (src, tag, tag))


# ---------------------------------------------------------------------------------------------------------------------------
def _stats(tag, name):
    p = os.path.join(PROF, "%s_%s.csv" % (tag, name))
    if not os.path.exists(p):
        return {}
    return {_kname(r["Name"]): r for r in csv.DictReader(open(p)) if "soccer::" in r["Name"]}


def _line(tag, name):
    p = os.path.join(PROF, "%s_%s.json" % (tag, name))
    return json.load(open(p)) if os.path.exists(p) else None


def _us(ns):
    return "%.2f" % (float(ns) / 1e3)


def _frac(us):
    return "%.3f" % (ALGO / (us * 1e-6) / PEAK)


def report(tag):
    L = []
    w = L.append
    own = os.path.join(PROF, "%s_traffic.json" % tag)           # (traffic.json itself moves on to the latest capture)
    t = json.load(open(own if os.path.exists(own) else os.path.join(PROF, "traffic.json")))
    step_prefix = "soccer::step_kernel_swar<0, 0, false, 1"     # (+ ", false>" since round 4's EXPL parameter)
    w("# %s — generated by `tools/collect_profile.py report %s` from the tracked files `profiles/%s_*`" % (tag, tag, tag))
    w("")
    w("Do not edit: every figure below is recomputed from those files (`tests/test_profiles_report.py` regenerates this file and compares).")
    w("Capture: `tools/profile_run.sh %s` on one MI355X (2^20 lanes, 5x4 pitch, slip 0 unless stated), folded by `tools/collect_profile.py fold`." % tag)
    if t.get("build") == tag:
        w("Kernel sources of the PMC passes: csrc sha256 `%s`, commit `%s`." % (t.get("csrc_sha256", "")[:16], t.get("commit")))
    w("Algorithmic bytes per step launch: 19 B x 2^20 = %d B (SURVEY.md 8(d)); peak 8 TB/s (MI355X_MICROARCH.md)." % ALGO)
    w("")
    # ---- 1. the step kernel's clocks -------------------------------------------------------------------------------------
    ks, kf, kd = _stats(tag, "kernel_stats"), _stats(tag, "kernel_stats_full"), _stats(tag, "kernel_stats_driver_shape")
    step = next((k for k in ks if k.startswith(step_prefix) and not k.endswith(", true>")), step_prefix + ">")
    un, ds, pr = _line(tag, "bench_unprofiled"), _line(tag, "bench_driver_shape"), _line(tag, "bench")
    sq = {}
    p = os.path.join(PROF, "%s_sq_counters.csv" % tag)
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            sq[(r["kernel"], r["counter"])] = r
    w("## 1. One launch of `%s`, four clocks, one build" % step)
    w("")
    w("| clock | us per launch | 19.92 MB / that / 8 TB/s | what it contains |")
    w("|---|---|---|---|")
    if (step, "SQ_BUSY_CYCLES") in sq:
        busy = float(sq[(step, "SQ_BUSY_CYCLES")]["mean_per_dispatch"]) / 32.0          # summed over the 32 shader engines
        wave = float(sq[(step, "SQ_WAVE_CYCLES")]["per_wave"]); wait = float(sq[(step, "SQ_WAIT_ANY")]["per_wave"])
        w("| SQ counters: `SQ_BUSY_CYCLES` %.0f per dispatch / 32 shader engines = %.0f cycles | %.2f at 2.4 GHz | %s | the span during which "
          "any wave of the launch is resident; one wave lives `SQ_WAVE_CYCLES` %.0f quad-cycles of which `SQ_WAIT_ANY` %.0f (%.0f %%) waiting |"
          % (busy * 32, busy, busy / 2400.0, _frac(busy / 2400.0), wave, wait, 100.0 * wait / wave))
    if step in ks:
        w("| rocprofv3 kernel trace, **minimum** duration (step-only command, %s calls) | %s | %s | begin-to-end of the quickest dispatch |"
          % (ks[step]["Calls"], _us(ks[step]["MinNs"]), _frac(float(ks[step]["MinNs"]) / 1e3)))
    if un:
        w("| device clock stamps around K = %d graph-replayed launches (`roofline.launch_us` of the un-profiled line) | %.2f | %.3f | start-to-start "
          "of consecutive dependent launches: execution + the dependent-launch boundary |" % (un["steps"], un["roofline"]["launch_us"], un["roofline"]["frac_device"]))
    if ds:
        w("| the same stamps at the driver's K = %d | %.2f | %.3f | + the replay's cold first launch |" % (ds["steps"], ds["roofline"]["launch_us"], ds["roofline"]["frac_device"]))
    for label, tab in (("step-only command", ks), ("default command", kf), ("driver-shape command", kd)):
        if step in tab:
            w("| rocprofv3 kernel trace, **average** duration (%s, %s calls) | %s | %s | under the profiler every dispatch is serialised through "
              "its own begin / end packets%s |" % (label, tab[step]["Calls"], _us(tab[step]["AverageNs"]), _frac(float(tab[step]["AverageNs"]) / 1e3),
                                                   "" if pr is None or label != "step-only command" else
                                                   ": the profiled run's own stamps read %.2f us per launch" % pr["roofline"]["launch_us"]))
    w("")
    w("Reading: the kernel-trace *average* is a property of the profiled run (dispatches serialised), the un-profiled device stamps are the cadence the")
    w("product runs at, and both lie between the SQ span / trace minimum (the kernel alone) below and the host wall clock of `value` above.")
    w("")
    # ---- 2. bench lines -----------------------------------------------------------------------------------------------------
    w("## 2. Bench lines of this build (un-profiled unless named)")
    w("")
    w("| line (`profiles/%s_<name>.json`) | K | value env-steps/s | ms_per_step | frac (wall) | frac_device | launch_us | host_overhead_us | regime | runtime |" % tag)
    w("|---|---|---|---|---|---|---|---|---|---|")
    for name in ("bench_unprofiled", "bench_driver_shape", "bench_driver_shape_again", "bench_driver_shape_torch_runtime", "bench_slip_unprofiled",
                 "bench_2rank_rehearsal", "bench", "bench_driver_shape_profiled"):
        d = _line(tag, name)
        if not d:
            continue
        r = d["roofline"]; host = d["config"].get("host", {})
        w("| `%s`%s | %d | %.3e | %.5f | %.3f | %.3f | %.2f | %.1f | %s | %s%s |"
          % (name, " (profiled)" if name in ("bench", "bench_driver_shape_profiled") else "", d["steps"], d["value"], d["ms_per_step"], r["frac"], r["frac_device"],
             r["launch_us"], r["host_overhead_us"], r.get("regime", "-"), host.get("hip_runtime", "torch"), ", slip %g" % d["config"]["slip_prob"] if d["config"]["slip_prob"] else ""))
    d2 = _line(tag, "bench_2rank_rehearsal")
    if d2 and d2.get("per_rank"):
        w("")
        w("Two-rank rehearsal on ONE GPU (`--gpus 2 --comm host`; the ranks share the device and take the timed region in turn, the exchange goes")
        w("through host files — never a measured path).  Per rank: " + "; ".join(
            "rank %d wall %.1f us, device region %.1f us, host %.1f us" % (x["rank"], x["wall_us"], x["device_region_us"], x["host_overhead_us"]) for x in d2["per_rank"])
          + ".  Gathered %d per-lane returns in global lane order." % d2["episodes"]["gathered_last_returns"])
    for name in ("bench_unprofiled", "bench_driver_shape"):
        d = _line(tag, name)
        if d and "fused_rollout" in d:
            v = d.get("vector_env_device") or {}
            w("")
            w("`%s`: fused rollout T = %d %.3e env-steps/s (%.3f of the HBM peak at %.2f B per env-step); config-5 self-play rollout %.3e; "
              % (name, d["fused_rollout"]["steps_fused"], d["fused_rollout"]["env_steps_per_s"], d["fused_rollout"]["frac_of_hbm_peak"],
                 d["fused_rollout"]["bytes_per_env_step"], d["selfplay_rollout_config5"]["env_steps_per_s"])
              + ("`VectorSoccerEnv(io='device')`: step() full %.2f us / lean %.2f us per call, rollout(T = %d) %.3e env-steps/s; "
                 % (v["full"]["us_per_step"], v["lean"]["us_per_step"], v["rollout"]["steps_fused"], v["rollout"]["env_steps_per_s"]) if "rollout" in v else "")
              + ("CPU oracle %.3e env-steps/s on %d threads (%.3e on one)." % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["value_1_core"])
                 if "cpu_baseline" in d else ""))
    w("")
    # ---- 3. kernel durations ------------------------------------------------------------------------------------------------
    w("## 3. Kernel durations (`rocprofv3 --kernel-trace --stats`; ns; profiled dispatches are serialised, see 1)")
    w("")
    w("| kernel | pass | calls | avg | min | max | bytes per launch | avg / min as a fraction of 8 TB/s |")
    w("|---|---|---|---|---|---|---|---|")
    others = _line(tag, "others") or {}
    bytes_of = [("step_kernel_swar<0,", 19 * N, "19 B/lane"), ("step_kernel_swar<1,", 23 * N, "23 B/lane (lean vector env)"),
                ("step_kernel_swar<2,", 31 * N, "31 B/lane (full vector env)"),
                ("reset_kernel_swar<false", 6 * N, "6 B/lane (+2 with the observation out)"), ("reset_kernel_swar<true", 15 * N, "1 + 6 + 6 + 2 B/lane"),
                ("trajectory_returns_kernel", None, "")]
    for name, lab in (("kernel_stats", "step-only"), ("kernel_stats_full", "default"), ("kernel_stats_slip0p2", "slip 0.2"),
                      ("kernel_stats_venv", "vector-env leg"), ("kernel_stats_other", "profile_others.py")):
        tab = _stats(tag, name)
        for k in sorted(tab):
            r = tab[k]
            nb, what = next(((b, wh) for pre, b, wh in bytes_of if pre in k.replace("soccer::", "")), (None, ""))
            if "step_kernel_swar<0" in k and k.endswith(", true>"):          # EXPL: + the two float64 uniform streams
                nb, what = 35 * N, "19 + 16 B/lane (caller-supplied uniforms)"
            if "rollout_swar_kernel<0," in k:                                 # action streams in, four trajectories out, state once
                bl = _line(tag, {"kernel_stats_full": "bench_full", "kernel_stats_slip0p2": "bench_slip_unprofiled", "kernel_stats_venv": "venv_profiled"}.get(name, ""))
                T = (bl.get("fused_rollout") or bl.get("rollout") or {}).get("steps_fused") if bl else None
                full = k.endswith(", true>")                                  # batched_rollout_ex: + final_obs (2 B) and prob_code (1 B) per step
                if T and name == "kernel_stats_venv" and not full:
                    T -= 1                                                    # rollout(infos="last"): T - 1 fused steps + one full step (infos="none": T)
                if T:
                    per = 10 if full else 7
                    nb, what = (per * T + 12) * N, "%d B/lane-step x %d + 12 B/lane" % (per, T)
            if "trajectory_returns_kernel" in k:                      # its T is the pass's: the bench line's K, or profile_others.py's 64
                bl = _line(tag, {"kernel_stats": "bench", "kernel_stats_full": "bench_full", "kernel_stats_slip0p2": "bench_slip_unprofiled"}.get(name, ""))
                T = bl["steps"] if bl else (64 if name == "kernel_stats_other" else None)
                if T:
                    nb, what = (3 * T + 5) * N, "3 B/lane-step x %d + 5 B/lane" % T
            if name == "kernel_stats_full" and k in _stats(tag, "kernel_stats"):
                continue
            fr = "%s / %s" % ("%.3f" % (nb / (float(r["AverageNs"]) * 1e-9) / PEAK), "%.3f" % (nb / (float(r["MinNs"]) * 1e-9) / PEAK)) if nb else ""
            w("| `%s` | %s | %s | %.0f | %s | %s | %s | %s |" % (k[:110], lab, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], what, fr))
    if others:
        w("")
        w("`profile_others.py` launched, at 2^20 lanes: " + "; ".join("%s x %d" % (k, v) for k, v in others.get("launches", {}).items()) + ".")
        w("(It re-uses the same buffers for all 100 launches of a kind: those working sets — 30 to 40 MB — stay in the 256 MB Infinity Cache, so a")
        w("fraction in its rows is bytes over time against the HBM peak, served from the cache, and can exceed 1.)")
    w("")
    # ---- 4. PMC traffic -----------------------------------------------------------------------------------------------------
    p = os.path.join(PROF, "%s_pmc_summary.csv" % tag)
    if os.path.exists(p):
        w("## 4. HBM-side traffic per launch (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950's")
        w("FETCH_SIZE reports half of a coalesced read stream — MI355X_MICROARCH.md, confirmed by the calibration copy kernel in `traffic.json`)")
        w("")
        w("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | HBM-side MB per launch |")
        w("|---|---|---|---|")
        m = collections.defaultdict(dict)
        for r in csv.DictReader(open(p)):
            m[r["kernel"]][r["counter"]] = float(r["mean_KB"])
        for k in sorted(m):
            if "FETCH_SIZE" in m[k] and "WRITE_SIZE" in m[k]:
                w("| `%s` | %.1f | %.1f | %.2f |" % (k[:100], m[k]["FETCH_SIZE"], m[k]["WRITE_SIZE"], (2 * m[k]["FETCH_SIZE"] + m[k]["WRITE_SIZE"]) * 1024 / 1e6))
        if t.get("build") == tag:
            w("")
            w("Step kernel: %.2f MB measured against %.2f MB algorithmic = **%.3f x** (`%s_traffic.json`; `traffic.json` = the latest capture's copy, what `bench.py` reports as `roofline.traffic`)."
              % (t["step_kernel_hbm_bytes_per_launch"] / 1e6, ALGO / 1e6, t["step_kernel_hbm_bytes_per_launch"] / ALGO, tag))
        w("")
    # ---- 5. SQ counters -----------------------------------------------------------------------------------------------------
    if sq:
        w("## 5. SQ counters per wave (`profiles/%s_sq_counters.csv`)" % tag)
        w("")
        names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"]
        w("| kernel | " + " | ".join(n.replace("SQ_", "") for n in names) + " |")
        w("|---|" + "---|" * len(names))
        for k in sorted({k for k, _ in sq}):
            cells = [sq[(k, n)]["per_wave"] if (k, n) in sq else "" for n in names]
            if any(cells):
                w("| `%s` | " % k[:100] + " | ".join(cells) + " |")
        w("")
    out = os.path.join(PROF, "%s_summary.md" % tag)
    text = "\n".join(L) + "\n"
    return out, text


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "fold":
        fold(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 3 and sys.argv[1] == "report":
        path, text = report(sys.argv[2])
        if len(sys.argv) > 3 and sys.argv[3] == "--stdout":
            sys.stdout.write(text)
        else:
            open(path, "w").write(text)
            print("wrote", os.path.relpath(path, ROOT))
    else:
        sys.exit(__doc__)
