!==============================================================================
! md_init_replay -- CPU-only companion of md_initial_config_gpu (test tool, no libljmd):
!   md_init_replay ran3 <seed> <count> <out.bin>
!       `count` draws of random_numbers::random_uniform as raw fp64 (stream file)
!   md_init_replay rv <epot.bin> <out rv_init.dat>
!       the init program's host arithmetic for inputs/input_simulation_parameters.txt with the
!       potential energy of the lattice read from epot.bin (one raw fp64, the value the force
!       routine returned at md_initial_config_program.f90:91) -- i.e. the hand-off file of a
!       warmup_steps = 0 run.  Fed the reference's own epot it must reproduce the reference's
!       rv_init.dat byte for byte.
!==============================================================================
program md_init_replay
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, sim_state, init_state
  use read_input_files, only: read_simulation_parameters
  use random_numbers,   only: random_uniform
  use md_init_host
  implicit none

  character(len=1024) :: mode, a1, a2, a3
  type(sim_params) :: params
  type(sim_state) :: state
  integer(kind=int_kind) :: total_steps, output_interval, warmup_steps, seed, count, i
  real(kind=dp_kind) :: rc_over_L, target_total_energy, epot, r
  integer :: iu, ios

  call get_command_argument(1, mode)
  call get_command_argument(2, a1)
  call get_command_argument(3, a2)
  call get_command_argument(4, a3)
  select case (trim(mode))
  case ('ran3')
    read(a1, *) seed
    read(a2, *) count
    open(newunit=iu, file=trim(a3), access='stream', form='unformatted', status='replace', action='write', iostat=ios)
    if (ios /= 0) stop 'md_init_replay: cannot open the output file'
    do i = 1, count
      r = random_uniform(seed)
      write(iu) r
    end do
    close(iu)
  case ('rv')
    call read_simulation_parameters('inputs/input_simulation_parameters.txt', params, total_steps, &
                                    output_interval, warmup_steps, rc_over_L, target_total_energy)
    call init_state(params, state)
    open(newunit=iu, file=trim(a1), access='stream', form='unformatted', status='old', action='read', iostat=ios)
    if (ios /= 0) stop 'md_init_replay: cannot open the epot file'
    read(iu) epot
    close(iu)
    call build_fcc_lattice(params, state)
    seed = -12345_int_kind
    call assign_random_velocities(params, state, seed)
    call remove_center_of_mass_velocity(params, state)
    call rescale_velocities_to_target_energy(params, state, target_total_energy, epot)
    call write_rv_init(trim(a2), params, state)
  case default
    stop 'usage: md_init_replay ran3 <seed> <count> <out.bin> | rv <epot.bin> <out.dat>'
  end select
end program md_init_replay
