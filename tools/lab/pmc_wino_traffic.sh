#!/bin/bash
# GPU lab: HBM-side fetch / write bytes of the Winograd kernels per shape and tile (FETCH_SIZE / WRITE_SIZE in separate passes): where the
# family's 2 x algorithmic READ traffic comes from -- the channel tiles of one window (Cout 256 = two tiles, Cout 128 = one) or the halo rows.
# usage: [LIBS="lab_wnx_1.so ..."] [SHAPES="..."] tools/lab/pmc_wino_traffic.sh   -> prints a table (-> profiles/); every library in LIBS (side builds under mcgaze_amd/,
#        tools/lab/wino_nt.sh) is measured beside the product library, with the plain launch time of each (no profiler) in a second table
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/wino_traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/[0-9]*
i=0
IFS='|' read -ra SHAPE_LIST <<< "${SHAPES:-448 56 56 256 256 6 randn 4|448 56 56 256 128 6 randn 4|448 56 56 256 256 6 randn 1|448 28 28 256 256 6 randn 4|448 14 14 256 256 6 relu 4|448 7 7 512 512 6 relu 4}"
for LIB in product $LIBS; do
  if [ "$LIB" != product ]; then export MCGAZE_LIB=$R/mcgaze_amd/$LIB; else unset MCGAZE_LIB; fi
  for ARGS in "${SHAPE_LIST[@]}"; do
    i=$((i+1))
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 150 rocprofv3 --kernel-trace --pmc $C -d $OUT/$i/$C -o $C --output-format csv -- python $R/tools/wino_bench.py $ARGS > $OUT/$i.$C.log 2>&1
    done
    echo "$LIB $ARGS" > $OUT/$i/args.txt
    A=($ARGS); python $R/tools/wino_bench.py ${A[0]} ${A[1]} ${A[2]} ${A[3]} ${A[4]} 40 ${A[6]} ${A[7]} 2>/dev/null | grep "^wino" | tail -1 > $OUT/$i/time.txt
  done
done
unset MCGAZE_LIB
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
print('| library | N H W Cin Cout .. tile | kernel | launches | fetch MB / launch (x 2: gfx950) | write MB / launch | algorithmic read / write MB | fetch / algorithmic | unprofiled run |')
print('|---|---|---|---|---|---|---|---|---|')
for d in sorted(glob.glob(f'{out}/[0-9]*'), key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 1 << 30):
    if not os.path.isdir(d): continue
    args = open(f'{d}/args.txt').read().split()
    lib, args = args[0], args[1:]
    N, H, W, Cin, Cout = map(int, args[:5])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in glob.glob(f'{d}/{sub}/**/*counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                if 'wino_x3' in row['Kernel_Name']:
                    agg[row['Kernel_Name'].split('(')[0].replace('void ', '')][sub].append(float(row['Counter_Value']))
    rd, wr = 4.0 * N * H * W * Cin / 1e6, 4.0 * N * H * W * Cout / 1e6
    for k, v in agg.items():
        f = sum(v['FETCH_SIZE']) / max(len(v['FETCH_SIZE']), 1) * 1024 * 2 / 1e6
        w = sum(v['WRITE_SIZE']) / max(len(v['WRITE_SIZE']), 1) * 1024 / 1e6
        tm = open(f'{d}/time.txt').read().strip() if os.path.exists(f'{d}/time.txt') else ''
        print(f"| {lib} | {' '.join(args)} | `{k}` | {len(v['FETCH_SIZE'])} | {f:.0f} | {w:.0f} | {rd:.0f} / {wr:.0f} | {f / rd:.2f} | {tm} |")
PY
