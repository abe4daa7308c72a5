#!/bin/bash
# several rocprofv3 --pmc passes over one bench step, summarised per kernel (k_wpair_dual first)
# usage: tools/pmc_passes.sh <tag>
tag=$1
B="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing"
tools/pmc_any.sh ${tag}_sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" $B | grep -A9 k_wpair_dual
tools/pmc_any.sh ${tag}_sq2 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU" $B | grep -A9 k_wpair_dual
tools/pmc_any.sh ${tag}_sq3 "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" $B | grep -A9 k_wpair_dual
tools/pmc_any.sh ${tag}_tcp1 "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" $B | grep -A5 k_wpair_dual
tools/pmc_any.sh ${tag}_tcp2 "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ" $B | grep -A5 k_wpair_dual
tools/pmc_any.sh ${tag}_tcp3 "TCP_GATE_EN1 TCP_GATE_EN2 TCP_TOTAL_CACHE_ACCESSES TCP_TA_TCP_STATE_READ" $B | grep -A5 k_wpair_dual
tools/pmc_any.sh ${tag}_tlb "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_STALL_INFLIGHT_MAX" $B | grep -A5 k_wpair_dual
tools/pmc_any.sh ${tag}_grbm "GRBM_GUI_ACTIVE GRBM_TA_BUSY" $B | grep -A3 k_wpair_dual
