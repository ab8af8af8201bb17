"""Copy the evidence of tools/collect_r03b.sh (gpurun_out/r03b, newest run directory of each profile) into profiles/ and rewrite the derived
figures in the headers of the three PMC summaries (formulas in the header lines themselves).  Run from the repo root after the GPU call."""
import glob, json, os, re, shutil
O = 'gpurun_out/r03b'
def newest(pat):
    fs = glob.glob(pat); fs.sort(key=os.path.getmtime); return fs[-1]
shutil.copy(f'{O}/bench.json', 'profiles/r03_bench.json')
shutil.copy(f'{O}/config_bench.jsonl', 'profiles/r03_config_bench.jsonl')
shutil.copy(newest(f'{O}/bench_prof/*/*_kernel_stats.csv'), 'profiles/r03_bench_kernel_stats.csv')
shutil.copy(newest(f'{O}/prof_cfg5/*/*_kernel_stats.csv'), 'profiles/r03_x5_cfg5_kernel_stats.csv')
shutil.copy(newest(f'{O}/prof_cfg3_train/*/*_kernel_stats.csv'), 'profiles/r03_cfg3_train_step_kernel_stats.csv')
def parse(fn):
    d = {}; n = 0
    for l in open(fn):
        m = re.match(r'(\S+)\s+launches=\s*(\d+)\s+per_launch=\s*(\d+)', l)
        if m: d[m.group(1)] = float(m.group(3)); n = int(m.group(2))
    return d, n
for src, dst, tj in (('s3_pmc_cfg2.txt', 'r03_s3_f16x3_d32_pmc_summary.txt', 'r03_s3_f16x3_d32_pmc_traffic.json'),
                     ('x3_pmc.txt', 'r03_x3_f16x3_d256_pmc_summary.txt', 'r03_x3_f16x3_d256_pmc_traffic.json'),
                     ('x5_pmc.txt', 'r03_x5_f16x3_d512_pmc_summary.txt', 'r03_x5_f16x3_d512_pmc_traffic.json')):
    d, n = parse(f'{O}/{src}')
    hdr = [l for l in open('profiles/' + dst).read().splitlines() if l.startswith('#')]
    cyc = d['GRBM_GUI_ACTIVE'] / 8 * 1024
    mp, va = d['SQ_VALU_MFMA_BUSY_CYCLES'] / cyc * 100, 4 * d['SQ_ACTIVE_INST_VALU'] / cyc * 100
    hbm = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024 / 1e6
    life = 4 * d['SQ_WAVE_CYCLES'] / d['SQ_WAVES']
    new = []
    for l in hdr:
        l = re.sub(r'average over (the )?\d+ launches', lambda m: f'average over {m.group(1) or ""}{n} launches', l)
        l = re.sub(r'(matrix pipe busy = .*= )[\d.]+ %', lambda m: m.group(1) + f'{mp:.1f} %', l)
        l = re.sub(r'(VALU busy = .*?= )[\d.]+ %', lambda m: m.group(1) + f'{va:.1f} %', l)
        l = re.sub(r'(HBM traffic per launch = .*= )[\d.]+ MB', lambda m: m.group(1) + f'{hbm:.1f} MB', l)
        if l.startswith('# against the life of a wave'):
            vb, mb = 4 * d['SQ_ACTIVE_INST_VALU'] / 1024, d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024
            l = (f'# against the life of a wave (4 x SQ_WAVE_CYCLES / SQ_WAVES = {life / 1e3:.1f} k cycles; excludes launch fill / drain): VALU busy 4 x SQ_ACTIVE_INST_VALU / 1024 = '
                 f'{vb / 1e3:.1f} k = {vb / life * 100:.0f} %, matrix pipe SQ_VALU_MFMA_BUSY_CYCLES / 1024 = {mb / 1e3:.1f} k = {mb / life * 100:.0f} %')
        new.append(l)
    open('profiles/' + dst, 'w').write('\n'.join(new + [l.rstrip('\n') for l in open(f'{O}/{src}')]) + '\n')
    t = json.load(open('profiles/' + tj))
    t['FETCH_SIZE_KB_per_launch'] = d['FETCH_SIZE']; t['WRITE_SIZE_KB_per_launch'] = d['WRITE_SIZE']; t['hbm_bytes_per_launch'] = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
    json.dump(t, open('profiles/' + tj, 'w'), indent=1)
    print(dst, f'matrix {mp:.1f} % valu {va:.1f} % hbm {hbm:.1f} MB n={n}')
b = json.load(open('profiles/r03_bench.json'))
print('value', b['value'], 'ms', b['ms_per_step'], 'roofline', b['roofline']['achieved'], b['roofline']['frac'], b['roofline'].get('kernel_ms_per_launch'))
print('train', b['train_step']['ms_per_step'], 'qgmm', b['value_with_query_gmm'], 'f32', b['f32']['ms_per_rollout'])
for k in ('d256', 'd512'):
    x = b[k]['f16x3']; print(k, x['ms_per_rollout'], x['roofline']['kernel_ms_per_launch'], x['roofline']['frac'], x.get('train_step'))
print(b['cpu_baseline']['value'], b['cpu_baseline']['train_step']['value'])
for l in open('profiles/r03_config_bench.jsonl'):
    r = json.loads(l); print(r.get('config'), round(r.get('ms_per_rollout', 0), 2), r.get('path'))
print(open(f'{O}/cfg3_train.log').read().split('cfg3 train step')[-1][:60])
