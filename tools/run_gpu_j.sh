#!/bin/sh
python tools/probes/overlap_probe.py 2>&1 | grep "fmt "
FLOAT_DEC_TPW=64 FLOAT_DEC_FLOW_WGS=256 python tools/probes/overlap_probe.py 2>&1 | grep "fmt "
FLOAT_DEC_TPW=256 FLOAT_DEC_FLOW_WGS=128 python tools/probes/overlap_probe.py 2>&1 | grep "fmt "
FLOAT_DEC_TPW=256 FLOAT_DEC_FLOW_WGS=128 FLOAT_DEC_ZBLUR_MIN=4096 python tools/probes/overlap_probe.py 2>&1 | grep "fmt "
