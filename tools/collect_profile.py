#!/usr/bin/env python3
"""Fold a gpurun_out/prof_X capture (bench.json + rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes, the
commands in profiles/r01_b_summary.md) into profiles/<tag>_{kernel_stats.csv,bench.json,pmc_summary.csv} and
profiles/traffic.json.  Usage: tools/collect_profile.py gpurun_out/prof_f r01_f"""
import collections, csv, glob, json, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ks = glob.glob(src + '/kt/runc/*_kernel_stats.csv')[0]
shutil.copy(ks, 'profiles/%s_kernel_stats.csv' % tag)
for r in csv.DictReader(open(ks)):
    if 'soccer::' in r['Name']:
        print("%-100s calls %5s avg %10.1f min %8s max %8s" % (r['Name'][:100], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
line = open(src + '/bench.json').read().strip().split('\n')[-1]
open('profiles/%s_bench.json' % tag, 'w').write(line + '\n')
d = json.loads(line)
print("bench: value %.4g launch_us %.3f frac %.4f rollout %.4g selfplay %.4g cpu %.3g" % (
    d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['fused_rollout']['env_steps_per_s'],
    d['selfplay_rollout_config5']['env_steps_per_s'], d['cpu_baseline']['value']))
out, means = [["run", "kernel", "counter", "dispatches", "mean_KB", "min_KB", "max_KB"]], {}
for kind, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(src + '/%s/runc/*_counter_collection.csv' % kind)[0])):
        if 'soccer::' in r['Kernel_Name']:
            name = re.sub(r'^void ', '', r['Kernel_Name']); name = name[:name.index('(')]
            agg[name].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        out.append(["bench", k, ctr, len(v), "%.1f" % (sum(v) / len(v)), "%.1f" % min(v), "%.1f" % max(v)])
        means[(k, ctr)] = sum(v) / len(v)
csv.writer(open('profiles/%s_pmc_summary.csv' % tag, 'w')).writerows(out)
t = json.load(open('profiles/traffic.json'))
k = [x for x in means if 'step_kernel_hot' in x[0]][0][0]
t.update(kernel=k, FETCH_SIZE_KB_mean=means[(k, 'FETCH_SIZE')], WRITE_SIZE_KB_mean=means[(k, 'WRITE_SIZE')],
         step_kernel_hbm_bytes_per_launch=(2 * means[(k, 'FETCH_SIZE')] + means[(k, 'WRITE_SIZE')]) * 1024)
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
print(open('profiles/%s_pmc_summary.csv' % tag).read())
print("traffic per launch: %.0f B" % t['step_kernel_hbm_bytes_per_launch'])
