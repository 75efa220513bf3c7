#!/bin/bash
# the driver's own command line, three times (fresh process each), and what the line says
for i in 1 2 3; do
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_$i.json 2> gpurun_out/driver_$i.err || { tail -5 gpurun_out/driver_$i.err; exit 1; }
  python3 tools/r4_show.py gpurun_out/driver_$i.json
done
