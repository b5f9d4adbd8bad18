"""profiles/<tag>_kernel_stats.md from rocprofv3 --kernel-trace --stats CSVs of bench.py (tools/, not product).
usage: kernel_stats.py <steps run> <overlap_stats.csv> <one_stream_stats.csv> > out.md"""
import collections
import csv
import re
import sys


def family(n):
    n = n.replace('void ', '')
    if n.startswith('(anonymous namespace)::'):
        n = n[len('(anonymous namespace)::'):]
    if n.startswith(('k_conv_x6', 'k_conv_h3r')):
        return 'conv fwd/dgrad: k_conv_x6'          # (label kept: the r4 / r5 tables are regenerated from their CSVs by a test; r6: + k_conv_h3r)
    if n.startswith('k_wgrad_x6'):
        return 'weight gradient: k_wgrad_x6'
    if n.startswith(('k_wgrad', 'k_stem_wgrad')):
        return 'weight gradient: fp32 kernels + k_wgrad_reduce + stem'
    if n.startswith(('k_bn', 'k_norm', 'k_stats', 'k_seg_meanvar')):
        return 'normalisation (k_bn*, k_norm*, k_stats*)'
    if n.startswith(('k_sum_pairs', 'k_sum_parts')):
        return 'fixed-order sums of split launches (k_sum_pairs / k_sum_parts)'
    if n.startswith('k_amax'):
        return 'amax passes of convolution operands (k_amax)'
    if n.startswith(('k_plan', 'k_radix')):
        return 'coordinates / maps'
    if n.startswith('k_x6_weight'):
        return 'weight images (k_x6_weight_image[s])'
    if n.startswith(('k_hash', 'k_unique', 'k_kernel_map', 'k_map_', 'k_gen_', 'k_union', 'k_child', 'k_nbr', 'k_permute_nbr', 'k_pairs_', 'k_scan',
                     'k_compact', 'k_voxel', 'k_morton', 'k_winner', 'k_flag', 'k_table', 'k_fill_i32', 'k_gather_coords', 'k_interp')):
        return 'coordinates / maps'
    if n.startswith('k_'):
        return 'other HIP kernels (stem, pool, head, loss, assign, optimizer, executor glue)'
    return 'torch (elementwise, cat, sort, copies, fills)'


def table(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    fam = collections.OrderedDict()
    for r in rows:
        d = fam.setdefault(family(r['Name']), [0, 0.0])
        d[0] += int(r['Calls'])
        d[1] += float(r['TotalDurationNs'])
    out = [f'Sum of kernel durations: {tot / 1e6 / steps:.2f} ms per step, {calls / steps:.0f} launches per step (over {steps} steps).', '',
           '| family | launches / step | ms / step | share |', '|---|---|---|---|']
    for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        out.append(f'| {f} | {c / steps:.1f} | {t / 1e6 / steps:.2f} | {100 * t / tot:.1f} % |')
    out += ['', 'Top kernels:', '', '| kernel | launches / step | ms / step | average us |', '|---|---|---|---|']
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:18]:
        name = re.split(r'\(', r['Name'].replace('void ', ''))[0]
        out.append(f"| `{name}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['AverageNs']) / 1e3:.1f} |")
    return '\n'.join(out)


if __name__ == '__main__':
    steps = float(sys.argv[1])
    print('### weight gradients + head branch on their streams (default)\n')
    print(table(sys.argv[2], steps))
    print('\n### `--no-wgrad-overlap` (one stream: a kernel\'s own time)\n')
    print(table(sys.argv[3], steps))
