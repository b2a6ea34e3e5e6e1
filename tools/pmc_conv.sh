cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmcq_*
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -c1-12 | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /root/repo/gpurun_out/pmcq_$tag -o r01 -- python /root/repo/tools/kbench.py ${KERNELS:-conv1_fwd conv1_dgrad conv1_wgrad} > /root/repo/gpurun_out/pmcq_$tag.log 2>&1
done
ls /root/repo/gpurun_out | grep pmcq
