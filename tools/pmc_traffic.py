"""HBM bytes per launch of one kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_*.sh, as bench.py's `roofline.traffic` reads it:
    python tools/pmc_traffic.py <pmc_dir> "<kernel needle>" <out.json> "<note>"
Units and the gfx950 correction as MI355X_MICROARCH.md prescribes: both counters are in KB; FETCH_SIZE counts half of the bytes of wide
(16 B / lane) coalesced reads, so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes, averaged over the launches of the passes."""
import collections, csv, glob, json, sys
root, needle, out, note = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "")
acc = collections.OrderedDict()
name = None
for f in sorted(glob.glob(root + '/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if needle in r['Kernel_Name'] and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE', 'GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_MFMA'):
            a = acc.setdefault(r['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
            name = r['Kernel_Name']
d = {k: v / n for k, (n, v) in acc.items()}
n = acc['FETCH_SIZE'][0]
cyc = d.get('GRBM_GUI_ACTIVE', 0) / 8 * 1024
res = {"kernel": name, "launches_averaged": n, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only; tools/collect_r04.sh)",
       "FETCH_SIZE_KB_per_launch": d['FETCH_SIZE'], "WRITE_SIZE_KB_per_launch": d['WRITE_SIZE'],
       "correction": "gfx950: FETCH_SIZE counts 1/2 of wide (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md, HBM)",
       "hbm_bytes_per_launch": (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024,
       "matrix_pipe_busy_pct": d['SQ_VALU_MFMA_BUSY_CYCLES'] / cyc * 100 if cyc and 'SQ_VALU_MFMA_BUSY_CYCLES' in d else None,
       "valu_busy_pct": 4 * d['SQ_ACTIVE_INST_VALU'] / cyc * 100 if cyc and 'SQ_ACTIVE_INST_VALU' in d else None,
       "SQ_INSTS_MFMA_per_launch": d.get('SQ_INSTS_MFMA'), "note": note}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
