O=gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
bash tools/session.sh r03n tests smoke bench stats
bash tools/pmc_session.sh r03n/pmc_main "f16x3 wino convB rfft irfft" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" > $O/pmc_main.log 2>&1
bash tools/pmc_session.sh r03n/pmc_conv1 "f16x3 conv1" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" > $O/pmc_conv1.log 2>&1
bash tools/pmc_session.sh r03n/pmc_fuconv "f16x3 fuconv" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" > $O/pmc_fuconv.log 2>&1
tail -5 $O/pmc_main.log
