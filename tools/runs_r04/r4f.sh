#!/bin/bash
out=gpurun_out/r4f; mkdir -p $out
timeout -k 10 300 python tools/runs_r04/scatter_probe2.py > $out/scatter_probe2.log 2>&1; cat $out/scatter_probe2.log
python tools/gpu_telemetry.py > $out/telemetry.json 2>&1; cat $out/telemetry.json
ls /sys/class/drm/ > $out/drm.txt 2>&1; ls /sys/class/drm/card*/device/ >> $out/drm.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/*/ >> $out/drm.txt 2>&1
(rocm-smi --showclocks --showpower --showtemp --json 2>&1 | head -c 3000) > $out/rocm_smi.txt
