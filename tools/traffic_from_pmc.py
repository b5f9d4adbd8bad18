"""profiles/<tag>_traffic.json from two rocprofv3 PMC passes over bench.py (one --pmc FETCH_SIZE, one --pmc WRITE_SIZE,
as MI355X_MICROARCH.md §HBM prescribes: separate passes; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 ->
x2, calibrated here with tools/pmc_calib.py: 1 GiB copy -> FETCH 0.5 GiB, WRITE 1.0 GiB).
usage: traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter, match):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and any(m in r['Kernel_Name'] for m in match):
            tot += float(r['Counter_Value'])
            n += 1
    return tot, n


def per_template(path, counter, match):
    """{kernel template: (sum of the counter, launches)} for the matching kernels"""
    out = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and any(m in r['Kernel_Name'] for m in match):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            d = out.setdefault(name, [0.0, 0])
            d[0] += float(r['Counter_Value'])
            d[1] += 1
    return out


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    match = ('k_conv_mfma', 'k_conv_glds', 'k_conv_x6', 'k_conv_h3r')        # the forward / backward-data family
    f, nf = per_kernel(fetch_csv, 'FETCH_SIZE', match)
    w, nw = per_kernel(write_csv, 'WRITE_SIZE', match)
    fetch_b = 2.0 * f * 1024 / max(nf, 1)        # KB -> B, x2 gfx950 correction
    write_b = w * 1024 / max(nw, 1)
    ft, wt = per_template(fetch_csv, 'FETCH_SIZE', match), per_template(write_csv, 'WRITE_SIZE', match)
    templates = {k: dict(launches=ft[k][1], fetch_MB_per_launch=round(2.0 * ft[k][0] * 1024 / ft[k][1] / 1e6, 2),
                         write_MB_per_launch=round(wt.get(k, [0.0, 1])[0] * 1024 / max(wt.get(k, [0.0, 1])[1], 1) / 1e6, 2)) for k in ft}
    from fcaf3d_amd.build import source_hash
    json.dump(dict(kernel_source_sha16=source_hash(), per_template=templates, kernel='k_conv_x6 + k_conv_h3r + k_conv_mfma* + k_conv_glds', launches_fetch_pass=nf, launches_write_pass=nw,
                   fetch_bytes_per_launch=round(fetch_b), write_bytes_per_launch=round(write_b),
                   hbm_bytes_per_launch=round(fetch_b + write_b),
                   method='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py; FETCH_SIZE x2 '
                          '(gfx950 under-count, calibrated on a 1 GiB copy: tools/pmc_calib.py)'),
              open(out, 'w'), indent=1)
    print(open(out).read())


if __name__ == '__main__':
    main()
