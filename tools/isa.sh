#!/bin/bash
# ISA of ONE sweep kernel: tools/isa.sh <n> [extra -D flags]  (n: 0 fwd, 1 bwd, 11 bwd_g, 15 bwd_lat_g, 7 bwd_x, 3 adj_bwd, 2 adj_fwd) -> /tmp/sdp_isa_<n>.s + a summary
N=${1:-1}; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ideepblast_amd/csrc -DSDP_ONLY=$N "$@" -S --cuda-device-only -o /tmp/sdp_isa_$N.s deepblast_amd/csrc/sdp_kernels.hip || exit 1
grep -E "^\s+\.(vgpr_count|sgpr_count|agpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):|NumVgprs|NumAgprs|ScratchSize|Occupancy" /tmp/sdp_isa_$N.s | sort | uniq -c | head -20
echo "instructions: $(grep -cE '^\s+(v_|s_|ds_|buffer_|global_|flat_)' /tmp/sdp_isa_$N.s)"
for pat in v_accvgpr s_waitcnt ds_read ds_write buffer_load buffer_store v_mov_b32_dpp s_sleep scratch_; do echo "$pat: $(grep -c "$pat" /tmp/sdp_isa_$N.s)"; done
