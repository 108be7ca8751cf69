!===============================================================================
! maniac_gpu -- ISO_C_BINDING view of the MI355X energy engine (include/maniac_gpu.h)
! plus reference-named wrappers for MANIAC's per-move energy seams.
!
! This is the module a MANIAC maintainer adds to src/ (see INTEGRATION.md).  The
! reference has no FFI: the hot path sits behind module procedures called from
! ComputeOldEnergy / ComputeNewEnergy (src/monte_carlo_utils.f90:275-395).  The
! wrappers below keep the reference's names, argument order and meaning
! (1-based residue type and molecule index, energies in Kelvin) and take the
! engine handle plus the arrays the reference keeps in module-global state
! (primary%mol_com, primary%site_offset), so energy_utils.f90 /
! ewald_energy.f90 bodies can be replaced one for one.
!
! Differences a caller must know:
!   * evaluation never mutates A(k): ComputeRecipEnergySingleMol returns the energy
!     of the trial state; GpuAcceptMove() applies it.  A rejected move needs no
!     Save/RestoreSingleMolFourier at all (ewald_phase.f90:134-255 become no-ops).
!   * status codes replace `stop` (output_utils.f90:535-562): every wrapper has an
!     optional `stat`; without it a failure calls `error stop` with the message.
!===============================================================================
module maniac_gpu

    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: real64

    implicit none

    integer(c_int), parameter :: MGPU_OK = 0
    integer(c_int), parameter :: MGPU_MOVE = 0, MGPU_CREATION = 1, MGPU_DELETION = 2, MGPU_NONE = 3
    integer(c_int), parameter :: MGPU_LANES = 4
    integer(c_int), parameter :: MGPU_FARM_DEPTH = 4                 ! farm windows a lane may have in flight (kFarmDepth)

    interface
        function mgpu_last_error() bind(C, name="mgpu_last_error") result(p)
            import :: c_ptr
            type(c_ptr) :: p
        end function
        function mgpu_engine_create(out, device, n_replicas, n_res, atoms_in_res, mol_capacity, max_atom, &
                                    atom_types, charges, is_active, n_types, epsilon, sigma, box_matrix, &
                                    bounds_lo, real_space_cutoff, ewald_tolerance) &
                                    bind(C, name="mgpu_engine_create") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), intent(out) :: out
            integer(c_int), value :: device, n_replicas, n_res, max_atom, n_types
            integer(c_int), intent(in) :: atoms_in_res(*), mol_capacity(*), atom_types(*), is_active(*)
            real(c_double), intent(in) :: charges(*), epsilon(*), sigma(*), box_matrix(9), bounds_lo(3)
            real(c_double), value :: real_space_cutoff, ewald_tolerance
            integer(c_int) :: rc
        end function
        function mgpu_engine_destroy(e) bind(C, name="mgpu_engine_destroy") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int) :: rc
        end function
        function mgpu_engine_get_ewald(e, alpha, rc_cut, tol, kmax, nk, volume, box_type) &
                bind(C, name="mgpu_engine_get_ewald") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            real(c_double), intent(out) :: alpha, rc_cut, tol, volume
            integer(c_int), intent(out) :: kmax(3), nk, box_type
            integer(c_int) :: rc
        end function
        function mgpu_replica_set_molecules(e, replica, t, n_mol, sites) &
                bind(C, name="mgpu_replica_set_molecules") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: replica, t, n_mol
            real(c_double), intent(in) :: sites(*)
            integer(c_int) :: rc
        end function
        function mgpu_replica_num_molecules(e, replica, t, n_mol) bind(C, name="mgpu_replica_num_molecules") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), value :: replica, t
            integer(c_int), intent(out) :: n_mol
            integer(c_int) :: rc
        end function
        function mgpu_replica_copy(e, dst, src) bind(C, name="mgpu_replica_copy") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), value :: dst, src
            integer(c_int) :: rc
        end function
        function mgpu_system_energy(e, replica, out) bind(C, name="mgpu_system_energy") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: replica
            real(c_double), intent(out) :: out(6)
            integer(c_int) :: rc
        end function
        subroutine mgpu_host_prefetch(p, bytes) bind(C, name="mgpu_host_prefetch")
            import :: c_ptr, c_int
            type(c_ptr), value :: p
            integer(c_int), value :: bytes
        end subroutine
        function mgpu_rng_seed_streams(seed, n_streams, state) bind(C, name="mgpu_rng_seed_streams") result(rc)
            import :: c_int, c_long_long
            integer(c_long_long), value :: seed
            integer(c_int), value :: n_streams
            integer(c_long_long), intent(out) :: state(*)
            integer(c_int) :: rc
        end function
        function mgpu_rng_fill(state, n_streams, n_per, u) bind(C, name="mgpu_rng_fill") result(rc)
            import :: c_int, c_long_long, c_double
            integer(c_long_long), intent(inout) :: state(*)
            integer(c_int), value :: n_streams, n_per
            real(c_double), intent(out) :: u(*)
            integer(c_int) :: rc
        end function
        function mgpu_replica_replace_molecule(e, replica, t, m_dst, m_src) &
                bind(C, name="mgpu_replica_replace_molecule") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), value :: replica, t, m_dst, m_src
            integer(c_int) :: rc
        end function
        function mgpu_replica_set_num_molecules(e, replica, t, n_mol) &
                bind(C, name="mgpu_replica_set_num_molecules") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), value :: replica, t, n_mol
            integer(c_int) :: rc
        end function
        function mgpu_structure_factor_add(e, replica, t, sites) bind(C, name="mgpu_structure_factor_add") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: replica, t
            real(c_double), intent(in) :: sites(*)
            integer(c_int) :: rc
        end function
        function mgpu_init_structure_factor(e, replica, mode) bind(C, name="mgpu_init_structure_factor") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), value :: replica, mode
            integer(c_int) :: rc
        end function
        function mgpu_pair_energy_candidates(e, n, replica, t, m, use_resident, sites, site_stride, e_nc, e_c) &
                bind(C, name="mgpu_pair_energy_candidates") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*), use_resident(*)
            real(c_double), intent(in) :: sites(*)
            real(c_double), intent(out) :: e_nc(*), e_c(*)
            integer(c_int) :: rc
        end function
        function mgpu_recip_energy_candidates(e, n, replica, t, m, kind, sites, site_stride, u) &
                bind(C, name="mgpu_recip_energy_candidates") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*), kind(*)
            real(c_double), intent(in) :: sites(*)
            real(c_double), intent(out) :: u(*)
            integer(c_int) :: rc
        end function
        function mgpu_self_energy(e, t, e_self) bind(C, name="mgpu_self_energy") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: t
            real(c_double), intent(out) :: e_self
            integer(c_int) :: rc
        end function
        function mgpu_intra_energy_candidates(e, n, replica, t, m, use_resident, sites, site_stride, u) &
                bind(C, name="mgpu_intra_energy_candidates") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*), use_resident(*)
            real(c_double), intent(in) :: sites(*)
            real(c_double), intent(out) :: u(*)
            integer(c_int) :: rc
        end function
        function mgpu_trial_energy_candidates(e, n, replica, t, m, sites, site_stride, old_energy, new_energy) &
                bind(C, name="mgpu_trial_energy_candidates") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*)
            real(c_double), intent(in) :: sites(*)
            real(c_double), intent(out) :: old_energy(*), new_energy(*)
            integer(c_int) :: rc
        end function
        function mgpu_commit_candidates(e, n, replica, t, m, kind, sites, site_stride, accept) &
                bind(C, name="mgpu_commit_candidates") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*), kind(*), accept(*)
            real(c_double), intent(in) :: sites(*)
            integer(c_int) :: rc
        end function
        function mgpu_trial_submit(e, lane, n, replica, t, m, sites, site_stride) &
                bind(C, name="mgpu_trial_submit") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane, n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*)
            real(c_double), intent(in) :: sites(*)
            integer(c_int) :: rc
        end function
        function mgpu_trial_wait(e, lane, old_energy, new_energy) bind(C, name="mgpu_trial_wait") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane
            real(c_double), intent(out) :: old_energy(*), new_energy(*)
            integer(c_int) :: rc
        end function
        ! sites: c_loc of a (3, site_stride, n) array, or c_null_ptr to commit the candidates of the
        ! lane's last mgpu_trial_submit, whose rows are still resident on the device
        function mgpu_commit_submit(e, lane, n, replica, t, m, kind, sites, site_stride, accept) &
                bind(C, name="mgpu_commit_submit") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e, sites
            integer(c_int), value :: lane, n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*), kind(*), accept(*)
            integer(c_int) :: rc
        end function
        ! molecule frames (com, offsets) resident on the device and trial moves built from them there
        function mgpu_replica_get_frames(e, replica, t, n_mol, com, off) bind(C, name="mgpu_replica_get_frames") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: replica, t
            integer(c_int), intent(out) :: n_mol
            real(c_double), intent(out) :: com(*), off(*)
            integer(c_int) :: rc
        end function
        function mgpu_move_trial_submit(e, lane, n, replica, t, m, move, u, translation_step, rotation_step) &
                bind(C, name="mgpu_move_trial_submit") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane, n
            integer(c_int), intent(in) :: replica(*), t(*), m(*), move(*)
            real(c_double), intent(in) :: u(*)
            real(c_double), value :: translation_step, rotation_step
            integer(c_int) :: rc
        end function
        ! the same trials with the acceptance test and the commit of accepted candidates on the device
        function mgpu_move_trial_decide_submit(e, lane, n, replica, t, m, move, u, translation_step, rotation_step, &
                                               accept_u, accept_pref, temperature) &
                bind(C, name="mgpu_move_trial_decide_submit") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane, n
            integer(c_int), intent(in) :: replica(*), t(*), m(*), move(*)
            real(c_double), intent(in) :: u(*), accept_u(*), accept_pref(*)
            real(c_double), value :: translation_step, rotation_step, temperature
            integer(c_int) :: rc
        end function
        function mgpu_trial_decide_wait(e, lane, old_energy, new_energy, accepted) bind(C, name="mgpu_trial_decide_wait") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane
            real(c_double), intent(out) :: old_energy(*), new_energy(*)
            integer(c_int), intent(out) :: accepted(*)
            integer(c_int) :: rc
        end function
        ! farm windows: one launch per lane step of a farm of chains (include/maniac_gpu.h)
        function mgpu_farm_window_capacity(e, max_chains, max_in_flight) bind(C, name="mgpu_farm_window_capacity") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), intent(out) :: max_chains, max_in_flight
            integer(c_int) :: rc
        end function
        function mgpu_farm_window_submit(e, lane, n, replica, t, m, move, forced, u5, accept_u, accept_pref, slot_u, &
                                         translation_step, rotation_step, temperature) &
                bind(C, name="mgpu_farm_window_submit") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane, n
            integer(c_int), intent(in) :: replica(*), t(*), m(*), move(*), forced(*)
            real(c_double), intent(in) :: u5(*), accept_u(*), accept_pref(*)
            type(c_ptr), value :: slot_u              ! c_null_ptr, or c_loc of n uniform numbers: the device picks the molecule
            real(c_double), value :: translation_step, rotation_step, temperature
            integer(c_int) :: rc
        end function
        function mgpu_farm_window_wait(e, lane, old_energy, new_energy, verdict) bind(C, name="mgpu_farm_window_wait") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane
            real(c_double), intent(out) :: old_energy(*), new_energy(*)
            integer(c_int), intent(out) :: verdict(*)
            integer(c_int) :: rc
        end function
        ! pinned staging of a lane's next trial: candidate rows built in place are not copied again
        function mgpu_lane_site_buffer(e, lane, n_max, site_stride, sites) bind(C, name="mgpu_lane_site_buffer") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), value :: lane, n_max, site_stride
            type(c_ptr), intent(out) :: sites
            integer(c_int) :: rc
        end function
        ! host threads the candidate loops inside submit / wait / commit may use (the calling driver's team)
        function mgpu_set_host_team(e, n_threads) bind(C, name="mgpu_set_host_team") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), value :: n_threads
            integer(c_int) :: rc
        end function
        ! mixed batches (moves, insertions, deletions); energies come back as rows of 5
        function mgpu_gcmc_trial_submit(e, lane, n, replica, t, m, kind, sites, site_stride) &
                bind(C, name="mgpu_gcmc_trial_submit") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane, n, site_stride
            integer(c_int), intent(in) :: replica(*), t(*), m(*), kind(*)
            real(c_double), intent(in) :: sites(*)
            integer(c_int) :: rc
        end function
        function mgpu_gcmc_trial_wait(e, lane, old_energy, new_energy) bind(C, name="mgpu_gcmc_trial_wait") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: lane
            real(c_double), intent(out) :: old_energy(*), new_energy(*)
            integer(c_int) :: rc
        end function
        ! one launch per window of trial steps of ONE chain: evaluate, decide in order, commit the first accepted step
        function mgpu_chain_window_capacity(e, max_candidates) bind(C, name="mgpu_chain_window_capacity") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int), intent(out) :: max_candidates
            integer(c_int) :: rc
        end function
        function mgpu_chain_window(e, replica, n, t, m, kind, link, sites, site_stride, accept_u, accept_pref, temperature, &
                                   recip_energy, old_energy, new_energy, first_accepted, undecided) &
                bind(C, name="mgpu_chain_window") result(rc)
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: e
            integer(c_int), value :: replica, n, site_stride
            integer(c_int), intent(in) :: t(*), m(*), kind(*), link(*)
            real(c_double), intent(in) :: sites(*), accept_u(*), accept_pref(*)
            real(c_double), value :: temperature, recip_energy
            real(c_double), intent(out) :: old_energy(*), new_energy(*)
            integer(c_int), intent(out) :: first_accepted, undecided
            integer(c_int) :: rc
        end function
        ! the path's one exchange step (RCCL all-gather of every rank's block sums and molecule-count histogram)
        function mgpu_comm_unique_id(id128) bind(C, name="mgpu_comm_unique_id") result(rc)
            import :: c_char, c_int
            character(kind=c_char), intent(out) :: id128(128)
            integer(c_int) :: rc
        end function
        function mgpu_comm_create(comm, device, rank, world, id128) bind(C, name="mgpu_comm_create") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), intent(out) :: comm
            integer(c_int), value :: device, rank, world
            type(c_ptr), value :: id128                     ! c_null_ptr for a single rank
            integer(c_int) :: rc
        end function
        function mgpu_comm_destroy(comm) bind(C, name="mgpu_comm_destroy") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: comm
            integer(c_int) :: rc
        end function
        function mgpu_allgather_block_stats(comm, n_sums, sums, n_bins, hist, sums_by_rank, hist_by_rank) &
                bind(C, name="mgpu_allgather_block_stats") result(rc)
            import :: c_ptr, c_int, c_double, c_long_long
            type(c_ptr), value :: comm
            integer(c_int), value :: n_sums, n_bins
            real(c_double), intent(in) :: sums(*)
            integer(c_long_long), intent(in) :: hist(*)
            real(c_double), intent(out) :: sums_by_rank(*)
            integer(c_long_long), intent(out) :: hist_by_rank(*)
            integer(c_int) :: rc
        end function
        function mgpu_synchronize(e) bind(C, name="mgpu_synchronize") result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: e
            integer(c_int) :: rc
        end function
        function c_strlen(s) bind(C, name="strlen") result(n)
            import :: c_ptr, c_size_t
            type(c_ptr), value :: s
            integer(c_size_t) :: n
        end function
    end interface

contains

    ! Message of the last failing engine call (mgpu_last_error), as a Fortran string
    function GpuLastError() result(msg)
        character(len=:), allocatable :: msg
        type(c_ptr) :: p
        character(kind=c_char), pointer :: s(:)
        integer :: n, i
        p = mgpu_last_error()
        n = int(c_strlen(p))
        allocate(character(len=n) :: msg)
        if (n > 0) then
            call c_f_pointer(p, s, [n])
            do i = 1, n
                msg(i:i) = s(i)
            end do
        end if
    end function GpuLastError

    ! status handling shared by the wrappers: AbortRun semantics unless `stat` is present
    subroutine GpuCheck(rc, where, stat)
        integer(c_int), intent(in) :: rc
        character(*), intent(in) :: where
        integer, intent(out), optional :: stat
        if (present(stat)) then
            stat = int(rc)
        else if (rc /= MGPU_OK) then
            write(*, '(A)') 'maniac_gpu: ' // where // ': ' // GpuLastError()
            error stop 1
        end if
    end subroutine GpuCheck

    ! absolute site coordinates com + offset of one molecule, as the reference forms them
    ! (geometry_utils.f90:379-382, ewald_phase.f90:398-399)
    pure subroutine MoleculeSites(mol_com, site_offset, natoms, sites)
        real(real64), intent(in) :: mol_com(3), site_offset(3, natoms)
        integer, intent(in) :: natoms
        real(real64), intent(out) :: sites(3, natoms)
        integer :: a
        do a = 1, natoms
            sites(:, a) = mol_com(:) + site_offset(:, a)
        end do
    end subroutine MoleculeSites

    !---------------------------------------------------------------------------
    ! ComputeSystemEnergy (energy_utils.f90:18-35): energy_out = non_coulomb, coulomb,
    ! recip_coulomb, ewald_self, intra_coulomb, total; then A(k) <- S(k), the initialisation
    ! the reference omits (SURVEY F2).
    !---------------------------------------------------------------------------
    subroutine ComputeSystemEnergy(engine, energy_out, stat)
        type(c_ptr), intent(in) :: engine
        real(real64), intent(out) :: energy_out(6)
        integer, intent(out), optional :: stat
        integer(c_int) :: rc
        rc = mgpu_system_energy(engine, 0_c_int, energy_out)
        if (rc == MGPU_OK) rc = mgpu_init_structure_factor(engine, 0_c_int, 1_c_int)
        call GpuCheck(rc, 'ComputeSystemEnergy', stat)
    end subroutine ComputeSystemEnergy

    !---------------------------------------------------------------------------
    ! ComputePairInteractionEnergy_singlemol (energy_utils.f90:374-442) for molecule
    ! (residue_type_1, molecule_index_1) placed at mol_com + site_offset.
    !---------------------------------------------------------------------------
    subroutine ComputePairInteractionEnergy_singlemol(engine, residue_type_1, molecule_index_1, mol_com, &
                                                      site_offset, natoms, e_non_coulomb, e_coulomb, stat)
        type(c_ptr), intent(in) :: engine
        integer, intent(in) :: residue_type_1, molecule_index_1, natoms
        real(real64), intent(in) :: mol_com(3), site_offset(3, natoms)
        real(real64), intent(out) :: e_non_coulomb, e_coulomb
        integer, intent(out), optional :: stat
        real(real64) :: sites(3, natoms), enc(1), ec(1)
        integer(c_int) :: rep(1), t(1), m(1), res(1), rc
        call MoleculeSites(mol_com, site_offset, natoms, sites)
        rep = 0; t = residue_type_1 - 1; m = molecule_index_1 - 1; res = 0
        rc = mgpu_pair_energy_candidates(engine, 1_c_int, rep, t, m, res, sites, int(natoms, c_int), enc, ec)
        e_non_coulomb = enc(1)
        e_coulomb = ec(1)
        call GpuCheck(rc, 'ComputePairInteractionEnergy_singlemol', stat)
    end subroutine ComputePairInteractionEnergy_singlemol

    !---------------------------------------------------------------------------
    ! SingleMolFourierTerms + ComputeRecipEnergySingleMol (ewald_phase.f90:383-420,
    ! ewald_energy.f90:191-274): reciprocal energy of the state in which the molecule sits at
    ! mol_com + site_offset (creation: is added; deletion: is removed).  A(k) is NOT mutated.
    !---------------------------------------------------------------------------
    subroutine ComputeRecipEnergySingleMol(engine, residue_type, molecule_index, mol_com, site_offset, natoms, &
                                           u_recipCoulomb_new, is_creation, is_deletion, stat)
        type(c_ptr), intent(in) :: engine
        integer, intent(in) :: residue_type, molecule_index, natoms
        real(real64), intent(in) :: mol_com(3), site_offset(3, natoms)
        real(real64), intent(out) :: u_recipCoulomb_new
        logical, intent(in), optional :: is_creation, is_deletion
        integer, intent(out), optional :: stat
        real(real64) :: sites(3, natoms), u(1)
        integer(c_int) :: rep(1), t(1), m(1), kind(1), rc
        call MoleculeSites(mol_com, site_offset, natoms, sites)
        rep = 0; t = residue_type - 1; m = molecule_index - 1; kind = MGPU_MOVE
        if (present(is_creation)) then
            if (is_creation) then
                kind = MGPU_CREATION
                m = -1
            end if
        end if
        if (present(is_deletion)) then
            if (is_deletion) kind = MGPU_DELETION
        end if
        rc = mgpu_recip_energy_candidates(engine, 1_c_int, rep, t, m, kind, sites, int(natoms, c_int), u)
        u_recipCoulomb_new = u(1)
        call GpuCheck(rc, 'ComputeRecipEnergySingleMol', stat)
    end subroutine ComputeRecipEnergySingleMol

    ! mc_acceptance_probability_swap (monte_carlo_utils.f90:228-268): acceptance of turning a molecule of a type with
    ! n_old molecules (fugacity phi_old) into one of a type with n_new molecules (phi_new); energies in K.  The
    ! reference has this rule and a `swap_proba` keyword but no swap move (monte_carlo.f90:50-75 has no branch for it).
    pure function mc_acceptance_probability_swap(old_total, new_total, n_old, n_new, phi_old, phi_new, temperature) &
            result(probability)
        real(real64), intent(in) :: old_total, new_total, phi_old, phi_new, temperature
        integer, intent(in) :: n_old, n_new
        real(real64) :: probability, combinatorial
        combinatorial = real(n_old, real64) / (real(n_new, real64) + 1.0_real64)
        probability = min(1.0_real64, (phi_new / phi_old) * combinatorial * exp(-(new_total - old_total) / temperature))
    end function mc_acceptance_probability_swap

    ! ComputeEwaldSelfInteractionSingleMol (ewald_energy.f90:308-336)
    subroutine ComputeEwaldSelfInteractionSingleMol(engine, residue_type, self_energy, stat)
        type(c_ptr), intent(in) :: engine
        integer, intent(in) :: residue_type
        real(real64), intent(out) :: self_energy
        integer, intent(out), optional :: stat
        integer(c_int) :: rc
        rc = mgpu_self_energy(engine, int(residue_type - 1, c_int), self_energy)
        call GpuCheck(rc, 'ComputeEwaldSelfInteractionSingleMol', stat)
    end subroutine ComputeEwaldSelfInteractionSingleMol

    ! ComputeIntraResidueRealCoulombEnergySingleMol (ewald_energy.f90:371-411)
    subroutine ComputeIntraResidueRealCoulombEnergySingleMol(engine, residue_type, molecule_index, mol_com, &
                                                             site_offset, natoms, u_intraCoulomb, stat)
        type(c_ptr), intent(in) :: engine
        integer, intent(in) :: residue_type, molecule_index, natoms
        real(real64), intent(in) :: mol_com(3), site_offset(3, natoms)
        real(real64), intent(out) :: u_intraCoulomb
        integer, intent(out), optional :: stat
        real(real64) :: sites(3, natoms), u(1)
        integer(c_int) :: rep(1), t(1), m(1), res(1), rc
        call MoleculeSites(mol_com, site_offset, natoms, sites)
        rep = 0; t = residue_type - 1; res = 0
        m = -1                             ! explicit sites are given: no resident slot is read
        if (molecule_index < 1) m = -1
        rc = mgpu_intra_energy_candidates(engine, 1_c_int, rep, t, m, res, sites, int(natoms, c_int), u)
        u_intraCoulomb = u(1)
        call GpuCheck(rc, 'ComputeIntraResidueRealCoulombEnergySingleMol', stat)
    end subroutine ComputeIntraResidueRealCoulombEnergySingleMol

    !---------------------------------------------------------------------------
    ! What AcceptMove / AcceptCreationMove / AcceptDeletionMove must additionally do
    ! (monte_carlo_utils.f90:410-422, create_molecule.f90:96-131, delete_molecule.f90:126-168):
    ! apply the accepted candidate to the engine (A(k) += delta, coordinates, counts).
    ! move_kind: MGPU_MOVE, MGPU_CREATION or MGPU_DELETION.
    !---------------------------------------------------------------------------
    subroutine GpuAcceptMove(engine, residue_type, molecule_index, move_kind, mol_com, site_offset, natoms, stat)
        type(c_ptr), intent(in) :: engine
        integer, intent(in) :: residue_type, molecule_index, move_kind, natoms
        real(real64), intent(in) :: mol_com(3), site_offset(3, natoms)
        integer, intent(out), optional :: stat
        real(real64) :: sites(3, natoms)
        integer(c_int) :: rep(1), t(1), m(1), kind(1), acc(1), rc
        call MoleculeSites(mol_com, site_offset, natoms, sites)
        rep = 0; t = residue_type - 1; m = molecule_index - 1; kind = int(move_kind, c_int); acc = 1
        if (move_kind == MGPU_CREATION) m = -1
        rc = mgpu_commit_candidates(engine, 1_c_int, rep, t, m, kind, sites, int(natoms, c_int), acc)
        call GpuCheck(rc, 'GpuAcceptMove', stat)
    end subroutine GpuAcceptMove

end module maniac_gpu
