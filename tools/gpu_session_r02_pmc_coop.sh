#!/bin/bash
# round 2: PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy, each its own run) over the cooperative local conv and, beside it, the 8-wave one
bash tools/pmc_session.sh r02pmc_coop "f16x3 convAc convA" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" 2>&1 | cut -c1-230
