#!/bin/bash
# round-2 GPU call M: intra-kernel timeline of small implicit-GEMM launches (tools/timeline_probe.py; the instrumented
# library tools/libhdu_tl.so is built beforehand with -DHDU_TIMELINE from the same sources)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/timeline_probe.py > gpurun_out/m_timeline.txt 2>&1
tail -80 gpurun_out/m_timeline.txt
