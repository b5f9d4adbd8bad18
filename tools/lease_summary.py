"""one-line digest of a bench.py JSON line (tools/lease.sh)"""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:                     # noqa
    print('no bench line:', e)
    sys.exit(0)
c, r = d['config'], d.get('roofline') or {}
h = c.get('host') or {}
print('   phases', h.get('phases_ms_per_step'), 'micro', h.get('micro'), 'pinned', h.get('pinned_to_gpu_numa_node'), 'settle', h.get('settle'))
print(f"value {d['value']} scenes/s  {d['ms_per_step']} ms/step  host_enqueue {h.get('host_enqueue_ms_per_step')} (busy {h.get('host_busy_ms_per_step')}, cpu {h.get('process_cpu_ms_per_step')} {h.get('thread_cpu_ms_per_step')})  load {((h.get('at_start') or {}).get('loadavg'))}"
      f"  roofline {r.get('achieved')} {r.get('frac')} {r.get('avg_launch_us')} us  sha {c.get('kernel_source_sha16')}")
print('   steps', h.get('steps'))
for k in ('fwd_bwd_only', 'forced_dp_n1', 'config4_per_gpu', 'bf16_fast_mode', 'literal_1cm', 'two_scales', 'sunrgbd', 's3dis', 'fp32_mfma_route',
          'inference', 'inference_pipelined'):
    v = c.get(k)
    if isinstance(v, dict):
        print('  ', k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'ms', 'scenes_per_s', 'ms_per_batch', 'error', 'host_enqueue_ms_per_step', 'host_busy_ms_per_step')},
              (v.get('calls') or {}).get('main_thread') or '', {kk: vv for kk, vv in (v.get('calls') or {}).items() if kk.startswith('call')}, {kk: vv for kk, vv in ((v.get('calls') or {}).get('kernel') or {}).items() if kk in ('minflt', 'device_mallocs', 'reserved_MB')})
for row in (r.get('hbm_kernels') or []):
    print('   hbm', row['kernel'], row['launches'], row['avg_us'], 'us', row['GBps'], 'GB/s', row['frac_of_hbm_peak'])
cb = d.get('cpu_baseline') or {}
print('   cpu', cb.get('value'), cb.get('cores'))
