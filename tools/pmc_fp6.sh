cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base nodma_noepi; do
  export SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/libspkdiff_$v.so
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc6/${v}_a -- python $R/tools/fp6_one.py 256 512 > $R/gpurun_out/pmc6/${v}_a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL --output-format csv -d $R/gpurun_out/pmc6/${v}_b -- python $R/tools/fp6_one.py 256 512 > $R/gpurun_out/pmc6/${v}_b.log 2>&1
done
ls -R $R/gpurun_out/pmc6 | head -30
