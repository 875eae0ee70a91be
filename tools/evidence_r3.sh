# Round-3 evidence run on the GPU box (one gpurun call): bench lines of all workloads, rocprofv3 kernel statistics of the
# config-3 step (single-stream and two-stream forms), the two PMC passes for HBM traffic, per-layer conv microbenchmark.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev_r3
mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_gan_x4.json 2> $O/bench_gan_x4.err
echo "bench done"; tail -n 3 $O/bench_gan_x4.err
for w in gen_l1_x4 infer_x8 dip_x2; do timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; tail -n 1 $O/bench_$w.err; done
timeout -k 10 300 python3 tools/microbench_conv.py > $O/microbench_conv.txt 2>&1
timeout -k 10 100 python3 tools/microbench_dense_adam.py > $O/microbench_dense_adam.txt 2>&1
timeout -k 10 100 python3 tools/microbench_first2.py > $O/microbench_first2.txt 2>&1
timeout -k 10 100 python3 tools/microbench_first2.py backward >> $O/microbench_first2.txt 2>&1
timeout -k 10 100 python3 tools/microbench_dgrad_bn.py > $O/microbench_dgrad_bn.txt 2>&1
timeout -k 10 100 python3 tools/microbench_dgrad_ps.py > $O/microbench_dgrad_ps.txt 2>&1
if [ -x tools/_bin/diag_dgrad_s2 ]; then
  { timeout -k 10 60 tools/_bin/diag_dgrad_s2 32 512 512 64 64 0; timeout -k 10 60 tools/_bin/diag_dgrad_s2 32 512 512 64 64 1; timeout -k 10 60 tools/_bin/diag_dgrad_s2 32 256 256 128 128 0; } > $O/diag_dgrad_s2.txt 2>&1
  if [ -x tools/_bin/diag_first2 ]; then { timeout -k 10 60 tools/_bin/diag_first2 1; timeout -k 10 60 tools/_bin/diag_first2 0; } > $O/diag_first2.txt 2>&1; fi
fi
timeout -k 10 100 python3 tools/microbench_first_bwd.py > $O/microbench_first_bwd.txt 2>&1
timeout -k 10 100 python3 tools/trace_step.py > $O/trace_step.txt 2>&1
cd /tmp && export TMPDIR=/tmp
DSR_GAN_OVERLAP=0 DSR_GAN_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o ser -- python3 $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-psnr --no-roofline --no-other-workloads > $O/prof_ser.log 2>&1
echo "prof serial done"
DSR_GAN_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_two -o two -- python3 $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-psnr --no-roofline --no-other-workloads > $O/prof_two.log 2>&1
echo "prof two-stream done"
for w in gen_l1_x4 infer_x8 dip_x2; do DSR_HIP_GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o k -- python3 $R/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-psnr --no-roofline --no-other-workloads > $O/prof_$w.log 2>&1; done
echo "prof small workloads done"
DSR_GAN_OVERLAP=0 DSR_GAN_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-roofline --no-other-workloads > $O/pmc_f.log 2>&1
echo "pmc fetch done"
DSR_GAN_OVERLAP=0 DSR_GAN_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-roofline --no-other-workloads > $O/pmc_w.log 2>&1
echo "pmc write done"
cd $R
python3 tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/traffic_gan_x4.json gan_x4
rm -rf $O/pmc_f $O/pmc_w
find $O -name "*kernel_trace.csv" -delete
ls $O
