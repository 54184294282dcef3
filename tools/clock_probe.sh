#!/bin/bash
# Effective clock under load and MFMA-pipe utilisation of the three MFMA kernels (MI355X_MICROARCH.md, DVFS give-back:
# effective clock = GRBM_GUI_ACTIVE / 8 / kernel time; the chip holds 1.5-2.0 GHz in MFMA-dense loops on random data).
# Two passes: kernel trace (durations) and PMC (separately: a profiled pass never carries both).
TAG=${1:-clk}
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt --output-format csv -- python3 $REPO/tools/clock_probe.py > $OUT/${TAG}_kt.txt 2> $OUT/${TAG}_kt.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/${TAG}_pmc -o pmc --output-format csv -- python3 $REPO/tools/clock_probe.py > $OUT/${TAG}_pmc.txt 2> $OUT/${TAG}_pmc.err
cd $REPO
python3 tools/summarize_prof.py $OUT/${TAG}_kt $OUT/${TAG}_pmc > $OUT/${TAG}_summary.txt
grep -E "scan_mfma|gemm_pp_kernel<4, 0>|gemm_pp|attention_dma|add_norm_kernel<3,0>|^#" $OUT/${TAG}_summary.txt | head -60
