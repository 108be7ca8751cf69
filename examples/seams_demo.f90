!===============================================================================
! seams_demo -- a stand-alone Fortran program that binds the engine the way INTEGRATION.md describes:
! it `use`s module maniac_gpu (ISO_C_BINDING interfaces + the reference-named wrappers), creates one engine
! from plain arrays, uploads a small SPC/E configuration read from a text file, and performs one
! translation trial of molecule `m` through the seams the reference's ComputeOldEnergy / ComputeNewEnergy
! call (src/monte_carlo_utils.f90:275-395), then accepts it.
!
!   amdflang -O2 examples/seams_demo.f90 -I<module dir> -L maniac_mc_amd -lmaniac_host -lmaniac_hip -o seams_demo
!   ./seams_demo config.txt
!
! config.txt:  n_mol L rc tol ; then n_mol lines "com(3) off(3,3)" ; then "m dx dy dz".
! Output: the six system energies, old and new (non_coulomb, coulomb, recip) of the trial, and the system
! energies after the accepted move -- all in Kelvin, 17 significant digits (tests/test_gpu_example.py
! compares them with the oracle).
!===============================================================================
program seams_demo

    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: real64
    use maniac_gpu

    implicit none

    type(c_ptr) :: engine
    integer(c_int) :: rc
    integer :: n_mol, m, i, a, u
    real(real64) :: box_len, rc_cut, tol, disp(3), e6(6), old(3), new(3)
    real(real64), allocatable :: com(:, :), off(:, :, :), sites(:, :, :)
    character(len=512) :: path
    ! SPC/E: one residue type, three sites, atom types 1 (O) and 2 (H); epsilon in K, sigma in Angstrom
    integer(c_int) :: atoms_in_res(1) = [3], capacity(1), atom_types(3) = [1, 2, 2], is_active(1) = [1]
    real(real64) :: charges(3) = [-0.8476_real64, 0.4238_real64, 0.4238_real64]
    real(real64) :: epsilon(4), sigma(4), box_matrix(9), bounds_lo(3)

    call get_command_argument(1, path)
    open(newunit=u, file=trim(path), status='old', action='read')
    read(u, *) n_mol, box_len, rc_cut, tol
    allocate(com(3, n_mol), off(3, 3, n_mol), sites(3, 3, n_mol))
    do i = 1, n_mol
        read(u, *) com(:, i), off(:, :, i)
    end do
    read(u, *) m, disp
    close(u)

    epsilon = [0.1553_real64 / 0.0019872041_real64, 0.0_real64, 0.0_real64, 0.0_real64]   ! kcal/mol -> K
    sigma = [3.166_real64, 0.0_real64, 0.0_real64, 0.0_real64]
    box_matrix = [box_len, 0.0_real64, 0.0_real64, 0.0_real64, box_len, 0.0_real64, 0.0_real64, 0.0_real64, box_len]
    bounds_lo = -0.5_real64 * box_len
    capacity = n_mol

    rc = mgpu_engine_create(engine, 0_c_int, 1_c_int, 1_c_int, atoms_in_res, capacity, 3_c_int, atom_types, charges, &
                            is_active, 2_c_int, epsilon, sigma, box_matrix, bounds_lo, rc_cut, tol)
    call GpuCheck(rc, 'mgpu_engine_create')
    do i = 1, n_mol
        do a = 1, 3
            sites(:, a, i) = com(:, i) + off(:, a, i)       ! as the reference forms them, geometry_utils.f90:379-382
        end do
    end do
    rc = mgpu_replica_set_molecules(engine, 0_c_int, 0_c_int, int(n_mol, c_int), sites)
    call GpuCheck(rc, 'mgpu_replica_set_molecules')

    call ComputeSystemEnergy(engine, e6)                     ! main.f90:27 (and A(k) <- S(k))
    write(*, '(A,6ES26.17)') 'system ', e6

    ! ComputeOldEnergy, default branch (monte_carlo_utils.f90:384-393)
    call ComputeRecipEnergySingleMol(engine, 1, m, com(:, m), off(:, :, m), 3, old(3))
    call ComputePairInteractionEnergy_singlemol(engine, 1, m, com(:, m), off(:, :, m), 3, old(1), old(2))
    ! the move (translation.f90:104-110 without the wrap: the demo keeps the displacement inside the box)
    com(:, m) = com(:, m) + disp
    ! ComputeNewEnergy, default branch (monte_carlo_utils.f90:310-318)
    call ComputeRecipEnergySingleMol(engine, 1, m, com(:, m), off(:, :, m), 3, new(3))
    call ComputePairInteractionEnergy_singlemol(engine, 1, m, com(:, m), off(:, :, m), 3, new(1), new(2))
    write(*, '(A,3ES26.17)') 'old    ', old
    write(*, '(A,3ES26.17)') 'new    ', new

    ! AcceptMove (monte_carlo_utils.f90:410-422) + the engine-side commit
    call GpuAcceptMove(engine, 1, m, int(MGPU_MOVE), com(:, m), off(:, :, m), 3)
    rc = mgpu_system_energy(engine, 0_c_int, e6)
    call GpuCheck(rc, 'mgpu_system_energy')
    write(*, '(A,6ES26.17)') 'after  ', e6
    rc = mgpu_engine_destroy(engine)

end program seams_demo
