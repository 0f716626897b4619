"""Regenerate profiles/ (round 3) from gpurun_out/r03 (profiles/scripts/profile_r03.sh + profile_r03b.sh) and gpurun_out/r03_eqv2."""
import csv, json, shutil, os
g = 'gpurun_out/r03/'
P = 'profiles/'
def rows(path):
    return list(csv.DictReader(open(path)))
def last_json(path):
    return json.loads([l for l in open(path) if l.startswith('{')][-1])
# kernel stats of the three bench commands
shutil.copy(g + 'stats/r03_kernel_stats.csv', P + 'r03_bench_kernel_stats.csv')
shutil.copy(g + 'stats/r03_domain_stats.csv', P + 'r03_bench_domain_stats.csv')
shutil.copy(g + 'eq_stats/eq_kernel_stats.csv', P + 'r03_eqv2_bench_kernel_stats.csv')
shutil.copy(g + 'train_stats/tr_kernel_stats.csv', P + 'r03_train_bench_kernel_stats.csv')
shutil.copy(g + 'message_launch_ms_by_step_and_layer.csv', P + 'r03_message_launch_ms_by_step_and_layer.csv')
for tag in ('pmc_fetch', 'pmc_write', 'pmc_util'):
    shutil.copy(g + tag + '_per_kernel.csv', P + 'r03_' + tag + '_per_kernel.csv')
    shutil.copy(g + 'eq_' + tag + '_per_kernel.csv', P + 'r03_eqv2_' + tag + '_per_kernel.csv')
# message kernel HBM traffic (FETCH x2 per the gfx950 correction, calibrated on the stream-copy kernel of the same runs)
f = {r['Kernel_Name']: r for r in rows(g + 'pmc_fetch_per_kernel.csv')}
w = {r['Kernel_Name']: r for r in rows(g + 'pmc_write_per_kernel.csv')}
k = [x for x in f if x.startswith('void adf_message_kernel<true, false')][0]
cal = [x for x in f if x.startswith('adf_peak_copy_kernel')][0]
fr = float(f[k]['FETCH_SIZE_mean_of_full_size_launches']) * 1024
wr = float(w[k]['WRITE_SIZE_mean_of_full_size_launches']) * 1024
d = last_json(g + 'bench.json')
N, H = 200000, 512
E = d['config']['edges_per_system'] * 1000
alg = (N + 1) * 5 * H * 4 + N * 4 * H * 4 + E * 20 + N * 4 * H * 4
out = {"kernel": "adf_message_kernel<f16x3, vec != 0>, full-size launches (all targets listed)", "systems": 1000,
       "fetch_size_bytes_raw": fr, "fetch_size_bytes_x2_gfx950": 2 * fr, "write_size_bytes": wr,
       "hbm_bytes_per_launch": 2 * fr + wr, "algorithmic_hbm_bytes_per_launch": alg,
       "calibration": {"kernel": "adf_peak_copy_kernel (1 GiB -> 1 GiB, 16 B per lane)",
                       "fetch_raw_bytes_per_launch": float(f[cal]['FETCH_SIZE_mean']) * 1024,
                       "write_bytes_per_launch": float(w[cal]['WRITE_SIZE_mean']) * 1024,
                       "note": "same runs: FETCH_SIZE reads 0.5x and WRITE_SIZE 1.0x of the known 2^30 bytes, confirming the gfx950 correction of MI355X_MICROARCH.md"},
       "note": "round 3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over `bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --num-steps 3` (profiles/r03_pmc_*_per_kernel.csv, counter unit KB); FETCH_SIZE doubled per MI355X_MICROARCH.md. Averages over the launches that evaluate every target. algorithmic = gather-record table + residual rows + 20 B per edge read, x_out/vec_out written, once each"}
json.dump(out, open(P + 'message_kernel_pmc.json', 'w'), indent=1)
d['roofline']['traffic'] = out['hbm_bytes_per_launch']
open(P + 'r03_bench.json', 'w').write(json.dumps(d) + '\n')
open(P + 'r03_bench_under_rocprof.json', 'w').write(json.dumps(last_json(g + 'under_rocprof.log')) + '\n')
open(P + 'r03_bench_train.json', 'w').write(json.dumps(last_json(g + 'bench_train.json')) + '\n')
# EquiformerV2: the dominant kernel's traffic from the 64-system PMC passes, per launch
ef = {r['Kernel_Name']: r for r in rows(g + 'eq_pmc_fetch_per_kernel.csv')}
ew = {r['Kernel_Name']: r for r in rows(g + 'eq_pmc_write_per_kernel.csv')}
e = last_json(g + 'bench_eqv2.json')
traffic = {}
for name in ef:
    if 'gemm16' in name:
        traffic[name[:60]] = {"launches": int(ef[name]['Launches']), "hbm_bytes_per_launch_64_systems":
                              2 * float(ef[name]['FETCH_SIZE_mean']) * 1024 + float(ew[name]['WRITE_SIZE_mean']) * 1024 if name in ew else None}
e['roofline']['traffic'] = traffic
e['roofline']['traffic_source'] = ("static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --model eqv2 --systems 64 --num-steps 2` "
                                   "(profiles/r03_eqv2_pmc_*_per_kernel.csv), 2*FETCH + WRITE per launch of the product kernels at 64 systems")
open(P + 'r03_bench_eqv2.json', 'w').write(json.dumps(e) + '\n')
if os.path.exists('gpurun_out/r03_eqv2/bench_eqv2_1000.json'):
    open(P + 'r03_bench_eqv2_1000_systems_earlier_build.json', 'w').write(json.dumps(last_json('gpurun_out/r03_eqv2/bench_eqv2_1000.json')) + '\n')
open(P + 'r03_bench_eqv2_under_rocprof.json', 'w').write(json.dumps(last_json(g + 'eq_under_rocprof.log')) + '\n')
print('message traffic/launch GB', out['hbm_bytes_per_launch'] / 1e9, 'alg', alg / 1e9)
print('painn', d['value'], 'under rocprof', last_json(g + 'under_rocprof.log')['value'])
for r in rows(P + 'r03_pmc_util_per_kernel.csv')[:10]: print(r['Kernel_Name'][:70], r['Launches'], r['MfmaUtil_mean'], r['VALUBusy_mean'])
print('eqv2', e['value'])
for r in rows(P + 'r03_eqv2_pmc_util_per_kernel.csv')[:10]: print(r['Kernel_Name'][:70], r['Launches'], r['MfmaUtil_mean'], r['VALUBusy_mean'])
ks = rows(P + 'r03_eqv2_bench_kernel_stats.csv'); tot = sum(float(r['TotalDurationNs']) for r in ks)
for r in sorted(ks, key=lambda r: -float(r['TotalDurationNs']))[:12]: print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us', round(100*float(r['TotalDurationNs'])/tot, 1), '%')
ks = rows(P + 'r03_bench_kernel_stats.csv'); tot = sum(float(r['TotalDurationNs']) for r in ks)
for r in sorted(ks, key=lambda r: -float(r['TotalDurationNs']))[:12]: print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us', round(100*float(r['TotalDurationNs'])/tot, 1), '%')
