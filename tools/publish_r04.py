"""Copy the evidence of tools/collect_r04.sh (gpurun_out/r04) into profiles/ under round-4 names and print the figures the docs quote.
Run from the repo root after the GPU call."""
import glob, json, os, shutil
O = 'gpurun_out/r04'
def newest(pat):
    fs = glob.glob(pat); fs.sort(key=os.path.getmtime); return fs[-1] if fs else None
def cp(src, dst):
    if src and os.path.exists(src):
        shutil.copy(src, 'profiles/' + dst); print('copied', dst)
    else:
        print('MISSING', src, '->', dst)
cp(f'{O}/bench.json', 'r04_bench.json')
cp(newest(f'{O}/bench_prof/*/*_kernel_stats.csv'), 'r04_bench_kernel_stats.csv')
cp(newest(f'{O}/prof_d256_train/*/*_kernel_stats.csv'), 'r04_d256_train_step_kernel_stats.csv')
cp(newest(f'{O}/prof_cfg3_train/*/*_kernel_stats.csv'), 'r04_cfg3_train_step_kernel_stats.csv')
cp(f'{O}/config_bench.jsonl', 'r04_config_bench.jsonl')
for t in ('s3_f16x3_d32', 'x3_f16x3_d256', 'x5_f16x3_d512'):
    cp(f'{O}/r04_{t}_pmc_traffic.json', f'r04_{t}_pmc_traffic.json')
for src, dst in (('s3_pmc_cfg2.txt', 'r04_s3_f16x3_d32_pmc_summary.txt'), ('x3_pmc.txt', 'r04_x3_f16x3_d256_pmc_summary.txt'), ('x5_pmc.txt', 'r04_x5_f16x3_d512_pmc_summary.txt')):
    cp(f'{O}/{src}', dst)
cp(f'{O}/ceiling.log', 'r04_mfma_ceiling_probe_run3.jsonl')
for f in ('cfg5_train.log', 'd256_train_time.log', 'cfg3_train.log', 'd256_train.log'):
    if os.path.exists(f'{O}/{f}'):
        print(f, open(f'{O}/{f}').read().strip().splitlines()[-2:])
try:
    b = json.load(open('profiles/r04_bench.json'))
    r = b['roofline']
    print('value %.4g' % b['value'], 'ms %.3f' % b['ms_per_step'], 'frac %.4f' % r['frac'], 'kernel_ms %.4f' % r['kernel_ms_per_launch'], 'traffic', r.get('traffic'),
          'prof_avg_us', r.get('profile_avg_launch_us'), 'frac_prof', r.get('frac_from_profile_avg'))
    print('train %.2f' % b['train_step']['ms_per_step'], 'qgmm %.4g' % b['value_with_query_gmm'], 'f32 %.2f' % b['f32']['ms_per_rollout'])
    for k in ('d256', 'd512'):
        x = b[k]['f16x3']; print(k, '%.2f ms' % x['ms_per_rollout'], 'kernel %.4f ms' % x['roofline']['kernel_ms_per_launch'], 'frac %.4f' % x['roofline']['frac'], 'traffic', x['roofline'].get('traffic'), x.get('train_step'))
    print('eig', {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ('frac', 'step_ms', 'history_ms', 'steps_ms', 'speedup')} for k, v in b['eig'].items() if isinstance(v, dict)})
    c = b['cpu_baseline']; print('cpu', c['value'], c.get('cores'), c.get('train_step', {}).get('value'))
except Exception as e:
    print('bench summary failed:', repr(e))
for t in ('s3_f16x3_d32', 'x3_f16x3_d256', 'x5_f16x3_d512'):
    try:
        j = json.load(open(f'profiles/r04_{t}_pmc_traffic.json'))
        print(t, 'hbm MB/launch %.1f' % (j['hbm_bytes_per_launch'] / 1e6), 'mfma busy %s' % j['matrix_pipe_busy_pct'], 'valu %s' % j['valu_busy_pct'], 'insts_mfma', j['SQ_INSTS_MFMA_per_launch'], 'n', j['launches_averaged'])
    except Exception as e:
        print(t, 'no traffic file', e)
