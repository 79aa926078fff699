#!/usr/bin/env python3
"""Fold a gpurun_out/prof_X capture (bench.json + rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes, the
commands in profiles/r01_b_summary.md) into profiles/<tag>_{kernel_stats.csv,bench.json,pmc_summary.csv} and
profiles/traffic.json.  Usage: tools/collect_profile.py gpurun_out/prof_f r01_f"""
import collections, csv, glob, json, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ks = (glob.glob(src + '/kt/runc/*_kernel_stats.csv') + glob.glob(src + '/kt/*kernel_stats.csv'))[0]
shutil.copy(ks, 'profiles/%s_kernel_stats.csv' % tag)
for r in csv.DictReader(open(ks)):
    if 'soccer::' in r['Name']:
        print("%-100s calls %5s avg %10.1f min %8s max %8s" % (r['Name'][:100], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
full = glob.glob(src + '/kt_full/runc/*_kernel_stats.csv') + glob.glob(src + '/kt_full/*kernel_stats.csv')
if full:
    shutil.copy(full[0], 'profiles/%s_kernel_stats_full.csv' % tag)
    print("-- default command (kt_full):")
    for r in csv.DictReader(open(full[0])):
        if 'soccer::' in r['Name']:
            print("%-100s calls %5s avg %10.1f min %8s max %8s" % (r['Name'][:100], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
    fl = open(src + '/bench_full.json').read().strip().split('\n')[-1]
    open('profiles/%s_bench_full.json' % tag, 'w').write(fl + '\n')
line = open(src + '/bench.json').read().strip().split('\n')[-1]
open('profiles/%s_bench.json' % tag, 'w').write(line + '\n')
d = json.loads(line)
print("bench (profiled run, kt): value %.4g launch_us %.3f frac %.4f" % (d['value'], d['roofline']['launch_us'], d['roofline']['frac']))
import os as _os
slipk = glob.glob(src + '/kt_slip/runc/*_kernel_stats.csv') + glob.glob(src + '/kt_slip/*kernel_stats.csv')
if slipk:
    shutil.copy(slipk[0], 'profiles/%s_kernel_stats_slip0p2.csv' % tag)
    print("-- slip 0.2 (kt_slip):")
    for r in csv.DictReader(open(slipk[0])):
        if 'soccer::' in r['Name']:
            print("%-100s calls %5s avg %10.1f min %8s max %8s" % (r['Name'][:100], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
k20 = glob.glob(src + '/kt_k20/runc/*_kernel_stats.csv') + glob.glob(src + '/kt_k20/*kernel_stats.csv')
if k20:
    shutil.copy(k20[0], 'profiles/%s_kernel_stats_driver_shape.csv' % tag)
    print("-- the driver's command, --steps 20 --warmup 5 (kt_k20):")
    for r in csv.DictReader(open(k20[0])):
        if 'soccer::' in r['Name']:
            print("%-100s calls %5s avg %10.1f min %8s max %8s" % (r['Name'][:100], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
for extra in ('bench_unprofiled.json', 'bench_driver_shape.json', 'bench_slip_unprofiled.json'):
    if _os.path.exists(src + '/' + extra):
        l2 = open(src + '/' + extra).read().strip().split('\n')[-1]
        open('profiles/%s_%s' % (tag, extra), 'w').write(l2 + '\n')
        e = json.loads(l2)
        print("%s: value %.4g launch_us %.3f frac %.4f" % (extra, e['value'], e['roofline']['launch_us'], e['roofline']['frac']))
out, means = [["run", "kernel", "counter", "dispatches", "mean_KB", "min_KB", "max_KB"]], {}
for kind, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open((glob.glob(src + '/%s/runc/*_counter_collection.csv' % kind) + glob.glob(src + '/%s/*counter_collection.csv' % kind))[0])):
        if 'soccer::' in r['Kernel_Name']:
            name = re.sub(r'^void ', '', r['Kernel_Name']); name = name[:name.index('(')]
            agg[name].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        out.append(["bench", k, ctr, len(v), "%.1f" % (sum(v) / len(v)), "%.1f" % min(v), "%.1f" % max(v)])
        means[(k, ctr)] = sum(v) / len(v)
csv.writer(open('profiles/%s_pmc_summary.csv' % tag, 'w')).writerows(out)
t = json.load(open('profiles/traffic.json'))
k = [x for x in means if 'step_kernel_swar' in x[0] or 'step_kernel_hot' in x[0]][0][0]
import hashlib, os, subprocess
def _csrc_sha():
    h = hashlib.sha256(); d = 'gym_soccer_littman94_amd/csrc'
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.hpp', 'Makefile')):
            h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()
t.update(round=int(tag[1:3]) if tag[:1] == 'r' and tag[1:3].isdigit() else t.get('round'), build=tag, csrc_sha256=_csrc_sha(), commit=subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip(),
         algorithmic_bytes_per_launch=19 * (1 << 20))
t.update(kernel=k, FETCH_SIZE_KB_mean=means[(k, 'FETCH_SIZE')], WRITE_SIZE_KB_mean=means[(k, 'WRITE_SIZE')],
         step_kernel_hbm_bytes_per_launch=(2 * means[(k, 'FETCH_SIZE')] + means[(k, 'WRITE_SIZE')]) * 1024)
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
print(open('profiles/%s_pmc_summary.csv' % tag).read())
print("traffic per launch: %.0f B" % t['step_kernel_hbm_bytes_per_launch'])

# SQ counters (two passes) -> profiles/<tag>_sq_counters.csv: mean per dispatch and per wave
rows = [["kernel", "counter", "dispatches", "mean_per_dispatch", "per_wave"]]
waves = {}
for sq in ('sq2', 'sq1', 'sq_slip'):
    files = glob.glob(src + '/%s/runc/*_counter_collection.csv' % sq) + glob.glob(src + '/%s/*counter_collection.csv' % sq)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if 'soccer::' in r['Kernel_Name']:
            name = re.sub(r'^void ', '', r['Kernel_Name']); name = name[:name.index('(')]
            agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()):
        if c == 'SQ_WAVES':
            waves[k] = sum(v) / len(v)
    for (k, c), v in sorted(agg.items()):
        m = sum(v) / len(v)
        rows.append([k, c, len(v), "%.0f" % m, "%.1f" % (m / waves[k]) if waves.get(k) else ""])
if len(rows) > 1:
    csv.writer(open('profiles/%s_sq_counters.csv' % tag, 'w')).writerows(rows)
    for r in rows:
        if 'step_kernel' in r[0]:
            print(r)
