#!/bin/bash
# BASELINE config 4 (N = 1 048 576 = 4 * 64^3) end to end through the thin Fortran drivers in ONE process each:
# md_initial_config_gpu (FCC + ran3 velocities + rescale, 2 warm-up steps) -> md_simulation_gpu with LJMD_GPUS=8.
# On a one-GPU box the eight ranks share the card (LJMD_DEVICES=0,...: peer-copy exchange); on an 8-GPU node drop
# LJMD_DEVICES and the exchange is RCCL over xGMI.  Measurement / demonstration tool: run through gpurun.
set -eo pipefail
PKG="$(cd "$(dirname "$0")/.." && pwd)/molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd"
W=$(mktemp -d)
mkdir -p "$W/inputs" "$W/outputs/one_run"
cat > "$W/inputs/input_simulation_parameters.txt" <<EOT
k total_steps output_interval warmup_steps
64 10 5 0
dt L rc_over_L
5.d-3 109.43846217182021d0 0.49d0
target_total_energy
-4.9d6
EOT
# warm-up steps of the init program come from the same file: 0 here (the lattice + rescaled velocities)
cd "$W"
"$PKG/bin/md_initial_config_gpu"
ls -la outputs/rv_init.dat
echo "== LJMD_GPUS=${LJMD_GPUS:-8} LJMD_DEVICES=${LJMD_DEVICES-0,0,0,0,0,0,0,0}"
LJMD_GPUS=${LJMD_GPUS:-8} LJMD_DEVICES=${LJMD_DEVICES-0,0,0,0,0,0,0,0} "$PKG/bin/md_simulation_gpu"
cat outputs/one_run/instantaneous_energies.dat
ls -la outputs/one_run/rva.dat
echo "== LJMD_GPUS=1"
LJMD_GPUS=1 "$PKG/bin/md_simulation_gpu" | tail -1
cat outputs/one_run/instantaneous_energies.dat
rm -rf "$W"
