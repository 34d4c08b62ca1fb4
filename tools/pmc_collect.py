"""Aggregate rocprofv3 --pmc counter_collection CSVs (one directory per pass) into per-kernel means.
    python tools/pmc_collect.py gpurun_out/pmc_r1f_* > profiles/r01_f_pmc.json
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
streaming reads (MI355X_MICROARCH.md, HBM section): fetch_bytes_corrected = FETCH_SIZE * 1024 * 2."""
import csv, glob, json, os, sys
from collections import defaultdict

KEEP = ('gemm_split_kernel', 'gemm_f32_mfma_kernel', 'attention_split_kernel', 'gmm_loglik_kernel', 'gmm_step1_kernel', 'gmm_step_kernel', 'layernorm_kernel',
        'gmm_operand_kernel', 'gmm_query_operand_kernel')
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        per_dispatch = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            key = next((k for k in KEEP if k in name), None)
            if key is None:
                continue
            if '<' in name.split(key, 1)[1][:4]:
                key = key + name.split(key, 1)[1].split('(')[0]
            dk = (key, r['Dispatch_Id'])
            per_dispatch[dk][r['Counter_Name']] = per_dispatch[dk].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            per_dispatch[dk]['_us'] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
        for (key, _), c in per_dispatch.items():
            for k, v in c.items():
                acc[key][k].append(v)
out = {}
for key, c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    o = {'launches_per_pass': len(c['_us']) // max(1, len(sys.argv) - 1), 'us_per_launch_under_pmc': round(m.pop('_us'), 1)}
    o.update({k: round(v, 1) for k, v in m.items()})
    if 'FETCH_SIZE' in m:
        o['fetch_bytes_corrected'] = m['FETCH_SIZE'] * 1024 * 2
        o['write_bytes'] = m.get('WRITE_SIZE', 0.0) * 1024
        o['traffic_bytes'] = o['fetch_bytes_corrected'] + o['write_bytes']
    if 'GRBM_GUI_ACTIVE' in m:
        o['effective_clock_GHz'] = round(m['GRBM_GUI_ACTIVE'] / 8 / o['us_per_launch_under_pmc'] / 1e3, 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        # busy cycles are summed over 256 CUs x 4 SIMDs; GUI_ACTIVE over 8 XCDs
        o['mfma_busy_frac'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 256 * 4), 4)
    if 'TCC_HIT_sum' in m:
        o['l2_hit_rate'] = round(m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']), 4)
    out[key] = o
print(json.dumps(out, indent=1))
