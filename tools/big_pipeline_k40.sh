set -e
W=$(mktemp -d); cd $W; mkdir -p inputs outputs/one_run
cat > inputs/input_simulation_parameters.txt <<EOT
# k total_steps output_interval warmup_steps
40 40 10 10
# dt L rc_over_L
5.d-3 68.39903786706788d0 0.49d0
# target energy
-1.2d6
EOT
P=/root/repo/molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd/bin
time $P/md_initial_config_gpu
ls -la outputs/rv_init.dat
time $P/md_simulation_gpu
cat outputs/one_run/instantaneous_energies.dat
head -24 outputs/one_run/md_final_results.txt
ls -la outputs/one_run/
