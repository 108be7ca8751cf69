!===============================================================================
! TEST INFRASTRUCTURE ONLY (oracle).  Never linked into, imported by, or called
! from the product path.
!
! bind(C) driver around the UNMODIFIED reference modules (compiled from
! /root/reference/src by oracle/Makefile into oracle/_ref/).  This file is our
! own code: it fills the reference's module-global state (simulation_state.f90)
! from flat C arrays -- replacing the file readers, which are out of scope for
! the hot path -- and then calls the reference's own hot-path routines, so that
! tests can obtain the reference's answers on arbitrary explicit inputs.
!
! Conventions: residue type / molecule / atom indices are 1-based exactly as in
! the reference.  Energies are in Kelvin (E/k_B), as in the reference.
!
! Known reference defects handled here (SURVEY F2/F3):
!   * ewald%recip_amplitude is never initialised by the reference
!     (prepare_utils.f90:228 allocates, ewald_energy.f90:244-255 only increments),
!     so ref_init_amplitude() must be called before any per-move recip routine.
!   * energy%ewald_self is accumulated without being zeroed
!     (energy_utils.f90:326); ref_system_energy() zeroes it first so that repeated
!     calls return what the first call returns.
!===============================================================================
module ref_shim

    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: real64
    use constants
    use parameters
    use simulation_state
    use helper_utils
    use geometry_utils
    use ewald_kvectors
    use ewald_phase
    use ewald_energy
    use energy_utils
    use prepare_utils
    use monte_carlo_utils
    use initoutput_utils
    use input_parser
    use data_parser
    use parameters_parser
    use tabulated_utils
    use montecarlo_module
    use random_utils

    implicit none

    logical, save :: is_setup = .false.
    character(len=200), save :: pending_res_file = ''

contains

    !---------------------------------------------------------------------------
    ! Release every allocatable the shim (or the reference) allocated.
    !---------------------------------------------------------------------------
    subroutine ref_teardown() bind(C, name="ref_teardown")
        if (allocated(primary%atom_charges)) deallocate(primary%atom_charges)
        if (allocated(primary%atom_types)) deallocate(primary%atom_types)
        if (allocated(primary%mol_com)) deallocate(primary%mol_com)
        if (allocated(primary%site_offset)) deallocate(primary%site_offset)
        if (allocated(primary%num_residues)) deallocate(primary%num_residues)
        if (allocated(coeff%sigma)) deallocate(coeff%sigma)
        if (allocated(coeff%epsilon)) deallocate(coeff%epsilon)
        if (allocated(input%fugacity)) deallocate(input%fugacity)
        if (allocated(input%is_active)) deallocate(input%is_active)
        if (allocated(nb%atom_in_residue)) deallocate(nb%atom_in_residue)
        if (allocated(res%site_offset_old)) deallocate(res%site_offset_old)
        if (allocated(ewald%recip_constants)) deallocate(ewald%recip_constants)
        if (allocated(ewald%recip_amplitude)) deallocate(ewald%recip_amplitude)
        if (allocated(ewald%recip_amplitude_old)) deallocate(ewald%recip_amplitude_old)
        if (allocated(ewald%form_factor)) deallocate(ewald%form_factor)
        if (allocated(ewald%phase_factor_x)) deallocate(ewald%phase_factor_x)
        if (allocated(ewald%phase_factor_y)) deallocate(ewald%phase_factor_y)
        if (allocated(ewald%phase_factor_z)) deallocate(ewald%phase_factor_z)
        if (allocated(ewald%phase_factor_x_old)) deallocate(ewald%phase_factor_x_old)
        if (allocated(ewald%phase_factor_y_old)) deallocate(ewald%phase_factor_y_old)
        if (allocated(ewald%phase_factor_z_old)) deallocate(ewald%phase_factor_z_old)
        if (allocated(ewald%temp_x)) deallocate(ewald%temp_x)
        if (allocated(ewald%temp_y)) deallocate(ewald%temp_y)
        if (allocated(ewald%temp_z)) deallocate(ewald%temp_z)
        if (allocated(ewald%phase_new)) deallocate(ewald%phase_new)
        if (allocated(ewald%phase_old)) deallocate(ewald%phase_old)
        if (allocated(ewald%charges)) deallocate(ewald%charges)
        if (allocated(ewald%kvectors)) deallocate(ewald%kvectors)
        is_setup = .false.
    end subroutine ref_teardown

    !---------------------------------------------------------------------------
    ! Define topology, force field and box, then run the reference's own setup:
    ! DetermineBoxSymmetry / ComputeCellProperties / ComputeInverse
    ! (what PrepareSimulationBox does minus logging, geometry_utils.f90:20-57),
    ! SetupEwald, AllocateArray, PrecomputeValidReciprocalVectors
    ! (what PrepareSimulationParameters does minus fugacity conversion and
    ! logging, prepare_utils.f90:19-41).
    !
    ! box_matrix : 9 doubles, column-major image of box%matrix(3,3)
    ! charges    : (n_res, max_atom) column-major, as primary%atom_charges
    ! atom_types : (n_res, max_atom) column-major, 1-based atom types
    ! eps, sig   : (n_atom_types, n_atom_types) in K and Angstrom
    ! Array shapes mirror AllocateAtomArrays (input_parser.f90:238-288).
    !---------------------------------------------------------------------------
    function ref_setup(n_res, atoms_in_res, max_atom, is_active, box_matrix, bounds_lo, &
                       is_triclinic, rc, tol, temp_K, charges, atom_types, n_atom_types, eps, sig) &
                       bind(C, name="ref_setup") result(rc_out)
        integer(c_int), value :: n_res, max_atom, n_atom_types, is_triclinic
        integer(c_int), intent(in) :: atoms_in_res(n_res), is_active(n_res)
        real(c_double), intent(in) :: box_matrix(9), bounds_lo(3)
        real(c_double), value :: rc, tol, temp_K
        real(c_double), intent(in) :: charges(n_res, max_atom)
        integer(c_int), intent(in) :: atom_types(n_res, max_atom)
        real(c_double), intent(in) :: eps(n_atom_types, n_atom_types), sig(n_atom_types, n_atom_types)
        integer(c_int) :: rc_out
        integer :: t1, t2, a1, a2, i

        call ref_teardown()

        nb%type_residue = n_res
        nb%max_atom_in_residue = max_atom
        allocate(nb%atom_in_residue(n_res))
        nb%atom_in_residue = atoms_in_res

        allocate(primary%atom_charges(n_res, max_atom))
        allocate(primary%atom_types(n_res, max_atom))
        allocate(primary%mol_com(3, n_res, NB_MAX_MOLECULE))
        allocate(primary%site_offset(3, n_res, NB_MAX_MOLECULE, max_atom))
        allocate(primary%num_residues(n_res))
        allocate(coeff%sigma(n_res, n_res, max_atom, max_atom))
        allocate(coeff%epsilon(n_res, n_res, max_atom, max_atom))
        allocate(input%fugacity(n_res))
        allocate(input%is_active(n_res))

        primary%atom_charges = charges
        primary%atom_types = atom_types
        primary%mol_com = zero
        primary%site_offset = zero
        primary%num_residues = 0
        input%is_active = is_active
        input%fugacity = one
        input%temp_K = temp_K
        input%real_space_cutoff = rc
        input%ewald_tolerance = tol

        ! epsilon/sigma are assigned purely by atom type in the reference
        ! (parameters_parser.f90:89-98, :141-176); replicate that mapping.
        coeff%sigma = zero
        coeff%epsilon = zero
        do t1 = 1, n_res
            do t2 = 1, n_res
                do a1 = 1, atoms_in_res(t1)
                    do a2 = 1, atoms_in_res(t2)
                        coeff%sigma(t1, t2, a1, a2) = sig(atom_types(t1, a1), atom_types(t2, a2))
                        coeff%epsilon(t1, t2, a1, a2) = eps(atom_types(t1, a1), atom_types(t2, a2))
                    end do
                end do
            end do
        end do

        primary%matrix = reshape(box_matrix, [3, 3])
        primary%bounds(:, 1) = bounds_lo
        do i = 1, 3
            primary%bounds(i, 2) = bounds_lo(i) + primary%matrix(i, i)
        end do
        primary%tilt = zero
        primary%is_triclinic = (is_triclinic /= 0)

        call DetermineBoxSymmetry(primary)
        call ComputeCellProperties(primary)
        call ComputeInverse(primary)

        call SetupEwald(verbose=.false.)
        call AllocateArray()
        call PrecomputeValidReciprocalVectors()

        energy%ewald_self = zero
        is_setup = .true.
        rc_out = 0
    end function ref_setup

    ! Fill all molecules of one residue type: com(3,n_mol), offsets(3,max_atom,n_mol)
    subroutine ref_set_molecules(res_type, n_mol, com, offsets) bind(C, name="ref_set_molecules")
        integer(c_int), value :: res_type, n_mol
        real(c_double), intent(in) :: com(3, n_mol)
        real(c_double), intent(in) :: offsets(3, nb%max_atom_in_residue, n_mol)
        integer :: m, a
        primary%num_residues(res_type) = n_mol
        do m = 1, n_mol
            primary%mol_com(:, res_type, m) = com(:, m)
            do a = 1, nb%max_atom_in_residue
                primary%site_offset(:, res_type, m, a) = offsets(:, a, m)
            end do
        end do
    end subroutine ref_set_molecules

    subroutine ref_set_num_residues(res_type, n_mol) bind(C, name="ref_set_num_residues")
        integer(c_int), value :: res_type, n_mol
        primary%num_residues(res_type) = n_mol
    end subroutine ref_set_num_residues

    function ref_get_num_residues(res_type) bind(C, name="ref_get_num_residues") result(n)
        integer(c_int), value :: res_type
        integer(c_int) :: n
        n = primary%num_residues(res_type)
    end function ref_get_num_residues

    subroutine ref_set_molecule(res_type, mol, com, offsets) bind(C, name="ref_set_molecule")
        integer(c_int), value :: res_type, mol
        real(c_double), intent(in) :: com(3)
        real(c_double), intent(in) :: offsets(3, nb%max_atom_in_residue)
        integer :: a
        primary%mol_com(:, res_type, mol) = com
        do a = 1, nb%max_atom_in_residue
            primary%site_offset(:, res_type, mol, a) = offsets(:, a)
        end do
    end subroutine ref_set_molecule

    subroutine ref_get_molecule(res_type, mol, com, offsets) bind(C, name="ref_get_molecule")
        integer(c_int), value :: res_type, mol
        real(c_double), intent(out) :: com(3)
        real(c_double), intent(out) :: offsets(3, nb%max_atom_in_residue)
        integer :: a
        com = primary%mol_com(:, res_type, mol)
        do a = 1, nb%max_atom_in_residue
            offsets(:, a) = primary%site_offset(:, res_type, mol, a)
        end do
    end subroutine ref_get_molecule

    ! Box products of the reference's own setup (geometry_utils.f90:68-154, :277-331)
    subroutine ref_get_box(box_type, volume, reciprocal, metrics) bind(C, name="ref_get_box")
        integer(c_int), intent(out) :: box_type
        real(c_double), intent(out) :: volume, reciprocal(9), metrics(9)
        box_type = primary%type
        volume = primary%volume
        reciprocal = reshape(primary%reciprocal, [9])
        metrics = primary%metrics
    end subroutine ref_get_box

    ! Ewald scalars from SetupEwald (prepare_utils.f90:103-214)
    subroutine ref_get_ewald(alpha, rc, tol, screening, fourier_precision, kmax, nk) bind(C, name="ref_get_ewald")
        real(c_double), intent(out) :: alpha, rc, tol, screening, fourier_precision
        integer(c_int), intent(out) :: kmax(3), nk
        alpha = ewald%alpha
        rc = input%real_space_cutoff
        tol = input%ewald_tolerance
        screening = ewald%screening_factor
        fourier_precision = ewald%fourier_precision
        kmax = ewald%kmax
        nk = ewald%num_kvectors
    end subroutine ref_get_ewald

    ! k-vector table (ewald_kvectors.f90:44-87) and W(k) (ewald_kvectors.f90:225-246)
    subroutine ref_get_kvectors(kx, ky, kz, k2norm, k2mag, form_factor, weights) bind(C, name="ref_get_kvectors")
        integer(c_int), intent(out) :: kx(ewald%num_kvectors), ky(ewald%num_kvectors), kz(ewald%num_kvectors)
        real(c_double), intent(out) :: k2norm(ewald%num_kvectors), k2mag(ewald%num_kvectors)
        real(c_double), intent(out) :: form_factor(ewald%num_kvectors), weights(ewald%num_kvectors)
        integer :: i
        call ComputeReciprocalWeights()
        do i = 1, ewald%num_kvectors
            kx(i) = ewald%kvectors(i)%kx
            ky(i) = ewald%kvectors(i)%ky
            kz(i) = ewald%kvectors(i)%kz
            k2norm(i) = ewald%kvectors(i)%k_squared
            k2mag(i) = ewald%kvectors(i)%k_squared_mag
            form_factor(i) = ewald%form_factor(i)
            weights(i) = ewald%recip_constants(i)
        end do
    end subroutine ref_get_kvectors

    ! ComputeSystemEnergy (energy_utils.f90:18-35).
    ! out = non_coulomb, coulomb, recip_coulomb, ewald_self, intra_coulomb, total
    subroutine ref_system_energy(out) bind(C, name="ref_system_energy")
        real(c_double), intent(out) :: out(6)
        energy%ewald_self = zero
        call ComputeSystemEnergy(primary)
        out(1) = energy%non_coulomb
        out(2) = energy%coulomb
        out(3) = energy%recip_coulomb
        out(4) = energy%ewald_self
        out(5) = energy%intra_coulomb
        out(6) = energy%total
    end subroutine ref_system_energy

    ! Build all per-atom phase tables (ewald_phase.f90:340-360) and W(k)
    subroutine ref_all_fourier_terms() bind(C, name="ref_all_fourier_terms")
        call ComputeReciprocalWeights()
        call ComputeAllFourierTerms()
    end subroutine ref_all_fourier_terms

    ! mode 0: A(k) <- 0 ; mode 1: A(k) <- full S(k) via the reference's own
    ! ComputeRecipAmplitude (ewald_energy.f90:40-77), using the current tables.
    subroutine ref_init_amplitude(mode) bind(C, name="ref_init_amplitude")
        integer(c_int), value :: mode
        integer :: i
        if (mode == 0) then
            ewald%recip_amplitude = (zero, zero)
        else
            do i = 1, ewald%num_kvectors
                ewald%recip_amplitude(i) = ComputeRecipAmplitude(ewald%kvectors(i)%kx, &
                    ewald%kvectors(i)%ky, ewald%kvectors(i)%kz)
            end do
        end if
    end subroutine ref_init_amplitude

    subroutine ref_get_amplitude(a) bind(C, name="ref_get_amplitude")
        real(c_double), intent(out) :: a(2, ewald%num_kvectors)
        integer :: i
        do i = 1, ewald%num_kvectors
            a(1, i) = real(ewald%recip_amplitude(i), kind=real64)
            a(2, i) = aimag(ewald%recip_amplitude(i))
        end do
    end subroutine ref_get_amplitude

    subroutine ref_set_amplitude(a) bind(C, name="ref_set_amplitude")
        real(c_double), intent(in) :: a(2, ewald%num_kvectors)
        integer :: i
        do i = 1, ewald%num_kvectors
            ewald%recip_amplitude(i) = cmplx(a(1, i), a(2, i), kind=real64)
        end do
    end subroutine ref_set_amplitude

    ! energy%recip_coulomb is what the creation/deletion "old" branches read
    ! (monte_carlo_utils.f90:372,381)
    subroutine ref_set_energy_recip(u) bind(C, name="ref_set_energy_recip")
        real(c_double), value :: u
        energy%recip_coulomb = u
    end subroutine ref_set_energy_recip

    ! ComputePairInteractionEnergy_singlemol (energy_utils.f90:374-442)
    subroutine ref_pair_singlemol(res_type, mol, e_non_coulomb, e_coulomb) bind(C, name="ref_pair_singlemol")
        integer(c_int), value :: res_type, mol
        real(c_double), intent(out) :: e_non_coulomb, e_coulomb
        call ComputePairInteractionEnergy_singlemol(primary, res_type, mol, e_non_coulomb, e_coulomb)
    end subroutine ref_pair_singlemol

    ! SingleMolPairwiseEnergy (energy_utils.f90:121-187), the ordered-pair variant
    subroutine ref_pair_ordered_singlemol(res_type, mol, e_non_coulomb, e_coulomb) &
            bind(C, name="ref_pair_ordered_singlemol")
        integer(c_int), value :: res_type, mol
        real(c_double), intent(out) :: e_non_coulomb, e_coulomb
        call SingleMolPairwiseEnergy(primary, res_type, mol, e_non_coulomb, e_coulomb)
    end subroutine ref_pair_ordered_singlemol

    ! ComputeDistance (geometry_utils.f90:359-415)
    function ref_distance(t1, m1, a1, t2, m2, a2) bind(C, name="ref_distance") result(d)
        integer(c_int), value :: t1, m1, a1, t2, m2, a2
        real(c_double) :: d
        d = ComputeDistance(primary, t1, m1, a1, t2, m2, a2)
    end function ref_distance

    ! ApplyPBC (geometry_utils.f90:167-220)
    subroutine ref_apply_pbc(pos) bind(C, name="ref_apply_pbc")
        real(c_double), intent(inout) :: pos(3)
        call ApplyPBC(pos, primary)
    end subroutine ref_apply_pbc

    ! LennardJonesEnergy / CoulombEnergy (energy_utils.f90:192-255)
    function ref_lj(r, sigma, eps) bind(C, name="ref_lj") result(e)
        real(c_double), value :: r, sigma, eps
        real(c_double) :: e
        e = LennardJonesEnergy(r, sigma, eps)
    end function ref_lj

    function ref_coulomb(r, q1, q2) bind(C, name="ref_coulomb") result(e)
        real(c_double), value :: r, q1, q2
        real(c_double) :: e
        e = CoulombEnergy(r, q1, q2)
    end function ref_coulomb

    ! SingleMolFourierTerms (ewald_phase.f90:383-420)
    subroutine ref_fourier_singlemol(res_type, mol) bind(C, name="ref_fourier_singlemol")
        integer(c_int), value :: res_type, mol
        call SingleMolFourierTerms(res_type, mol)
    end subroutine ref_fourier_singlemol

    ! Save / Restore / Replace (ewald_phase.f90:134-322)
    subroutine ref_save_fourier(res_type, mol) bind(C, name="ref_save_fourier")
        integer(c_int), value :: res_type, mol
        call SaveSingleMolFourierTerms(res_type, mol)
    end subroutine ref_save_fourier

    subroutine ref_restore_fourier(res_type, mol) bind(C, name="ref_restore_fourier")
        integer(c_int), value :: res_type, mol
        call RestoreSingleMolFourier(res_type, mol)
    end subroutine ref_restore_fourier

    subroutine ref_replace_fourier(res_type, index_1, index_2) bind(C, name="ref_replace_fourier")
        integer(c_int), value :: res_type, index_1, index_2
        call ReplaceFourierTermsSingleMol(res_type, index_1, index_2)
    end subroutine ref_replace_fourier

    ! One atom's three 1-D phase tables, k = -kmax..kmax, as (re,im) pairs
    subroutine ref_get_phase_tables(res_type, mol, atom, px, py, pz) bind(C, name="ref_get_phase_tables")
        integer(c_int), value :: res_type, mol, atom
        real(c_double), intent(out) :: px(2, -ewald%kmax(1):ewald%kmax(1))
        real(c_double), intent(out) :: py(2, -ewald%kmax(2):ewald%kmax(2))
        real(c_double), intent(out) :: pz(2, -ewald%kmax(3):ewald%kmax(3))
        integer :: k
        do k = -ewald%kmax(1), ewald%kmax(1)
            px(1, k) = real(ewald%phase_factor_x(res_type, mol, atom, k), kind=real64)
            px(2, k) = aimag(ewald%phase_factor_x(res_type, mol, atom, k))
        end do
        do k = -ewald%kmax(2), ewald%kmax(2)
            py(1, k) = real(ewald%phase_factor_y(res_type, mol, atom, k), kind=real64)
            py(2, k) = aimag(ewald%phase_factor_y(res_type, mol, atom, k))
        end do
        do k = -ewald%kmax(3), ewald%kmax(3)
            pz(1, k) = real(ewald%phase_factor_z(res_type, mol, atom, k), kind=real64)
            pz(2, k) = aimag(ewald%phase_factor_z(res_type, mol, atom, k))
        end do
    end subroutine ref_get_phase_tables

    ! ComputeRecipEnergySingleMol (ewald_energy.f90:191-274).
    ! mode 0 = move (new - old), 1 = creation (+new), 2 = deletion (-old).
    ! Mutates ewald%recip_amplitude exactly as the reference does.
    subroutine ref_recip_singlemol(res_type, mol, mode, u) bind(C, name="ref_recip_singlemol")
        integer(c_int), value :: res_type, mol, mode
        real(c_double), intent(out) :: u
        if (mode == 1) then
            call ComputeRecipEnergySingleMol(res_type, mol, u, is_creation=.true.)
        else if (mode == 2) then
            call ComputeRecipEnergySingleMol(res_type, mol, u, is_deletion=.true.)
        else
            call ComputeRecipEnergySingleMol(res_type, mol, u)
        end if
    end subroutine ref_recip_singlemol

    ! ComputeReciprocalEnergy (ewald_energy.f90:105-147) on the current tables
    subroutine ref_recip_total(u) bind(C, name="ref_recip_total")
        real(c_double), intent(out) :: u
        call ComputeReciprocalEnergy(u)
    end subroutine ref_recip_total

    ! ComputeEwaldSelfInteractionSingleMol (ewald_energy.f90:308-336)
    subroutine ref_self_singlemol(res_type, e) bind(C, name="ref_self_singlemol")
        integer(c_int), value :: res_type
        real(c_double), intent(out) :: e
        call ComputeEwaldSelfInteractionSingleMol(res_type, e)
    end subroutine ref_self_singlemol

    ! ComputeIntraResidueRealCoulombEnergySingleMol (ewald_energy.f90:371-411)
    subroutine ref_intra_singlemol(res_type, mol, e) bind(C, name="ref_intra_singlemol")
        integer(c_int), value :: res_type, mol
        real(c_double), intent(out) :: e
        call ComputeIntraResidueRealCoulombEnergySingleMol(res_type, mol, e)
    end subroutine ref_intra_singlemol

    ! ComputeOldEnergy / ComputeNewEnergy (monte_carlo_utils.f90:275-395).
    ! kind 0 = translation/rotation, 1 = creation, 2 = deletion.
    ! out = non_coulomb, coulomb, recip_coulomb, ewald_self, intra_coulomb, total.
    ! For kind 0 the reference leaves ewald_self / intra_coulomb unset
    ! (intent(out) struct, fields never assigned): reported as 0 here.
    subroutine ref_old_energy(res_type, mol, kind, out) bind(C, name="ref_old_energy")
        integer(c_int), value :: res_type, mol, kind
        real(c_double), intent(out) :: out(6)
        type(energy_state) :: e
        e%ewald_self = zero
        e%intra_coulomb = zero
        if (kind == 1) then
            call ComputeOldEnergy(res_type, mol, e, is_creation=.true.)
        else if (kind == 2) then
            call ComputeOldEnergy(res_type, mol, e, is_deletion=.true.)
        else
            call ComputeOldEnergy(res_type, mol, e)
            e%ewald_self = zero
            e%intra_coulomb = zero
        end if
        out = [e%non_coulomb, e%coulomb, e%recip_coulomb, e%ewald_self, e%intra_coulomb, e%total]
    end subroutine ref_old_energy

    subroutine ref_new_energy(res_type, mol, kind, out) bind(C, name="ref_new_energy")
        integer(c_int), value :: res_type, mol, kind
        real(c_double), intent(out) :: out(6)
        type(energy_state) :: e
        e%ewald_self = zero
        e%intra_coulomb = zero
        if (kind == 1) then
            call ComputeNewEnergy(res_type, mol, e, is_creation=.true.)
        else if (kind == 2) then
            call ComputeNewEnergy(res_type, mol, e, is_deletion=.true.)
        else
            call ComputeNewEnergy(res_type, mol, e)
            e%ewald_self = zero
            e%intra_coulomb = zero
        end if
        out = [e%non_coulomb, e%coulomb, e%recip_coulomb, e%ewald_self, e%intra_coulomb, e%total]
    end subroutine ref_new_energy

    ! mc_acceptance_probability (monte_carlo_utils.f90:184-226).
    ! move_type uses the reference ids (parameters.f90:36-39): 1 creation,
    ! 2 deletion, 3 translation, 4 rotation.  fugacity is the already-converted
    ! value (molecules per cubic Angstrom, prepare_utils.f90:68).
    function ref_acceptance(old_total, new_total, res_type, move_type, fugacity) &
            bind(C, name="ref_acceptance") result(p)
        real(c_double), value :: old_total, new_total, fugacity
        integer(c_int), value :: res_type, move_type
        real(c_double) :: p
        type(energy_state) :: eo, en
        eo%total = old_total
        en%total = new_total
        input%fugacity(res_type) = fugacity
        p = mc_acceptance_probability(eo, en, res_type, move_type)
    end function ref_acceptance

    ! mc_acceptance_probability_swap (monte_carlo_utils.f90:228-268): the only piece of a swap move the reference has
    function ref_acceptance_swap(old_total, new_total, type_old, type_new, fug_old, fug_new) &
            bind(C, name="ref_acceptance_swap") result(p)
        real(c_double), value :: old_total, new_total, fug_old, fug_new
        integer(c_int), value :: type_old, type_new
        real(c_double) :: p
        type(energy_state) :: eo, en
        eo%total = old_total
        en%total = new_total
        input%fugacity(type_old) = fug_old
        input%fugacity(type_new) = fug_new
        p = mc_acceptance_probability_swap(eo, en, type_old, type_new)
    end function ref_acceptance_swap

    ! PrecomputeTable + LookupTabulated (tabulated_utils.f90:10-113): the reference's tabulated potentials.  They are
    ! dead code in the reference (use_table is a compile-time .false., parameters.f90:42); the routines themselves
    ! are callable.  which: 1 erfc(alpha r)/r, 2 r**6, 3 r**12.  (Re)builds the tables on first use after a setup.
    function ref_table_lookup(which, r) bind(C, name="ref_table_lookup") result(f)
        integer(c_int), value :: which
        real(c_double), value :: r
        real(c_double) :: f
        f = 0.0_c_double
        if (.not. is_setup) return
        if (allocated(erfc_r_table%x)) then
            if (erfc_r_table%dx /= input%real_space_cutoff / real(TABULATED_POINTS, real64)) then
                deallocate(erfc_r_table%x, erfc_r_table%f, r6_table%x, r6_table%f, r12_table%x, r12_table%f)
            end if
        end if
        if (.not. allocated(erfc_r_table%x)) call PrecomputeTable()
        select case (which)
        case (1)
            f = LookupTabulated(erfc_r_table, r)
        case (2)
            f = LookupTabulated(r6_table, r)
        case default
            f = LookupTabulated(r12_table, r)
        end select
    end function ref_table_lookup

    ! RotationMatrix (helper_utils.f90:39-77), column-major 3x3 out
    subroutine ref_rotation_matrix(axis, theta, r) bind(C, name="ref_rotation_matrix")
        integer(c_int), value :: axis
        real(c_double), value :: theta
        real(c_double), intent(out) :: r(9)
        r = reshape(RotationMatrix(axis, theta), [9])
    end subroutine ref_rotation_matrix

    ! ConvertFugacity (prepare_utils.f90:48-73): atm -> molecules / A^3, run through
    ! the reference routine itself on residue type 1 (state saved and restored).
    function ref_convert_fugacity(f_atm, temp_K) bind(C, name="ref_convert_fugacity") result(f)
        real(c_double), value :: f_atm, temp_K
        real(c_double) :: f
        real(real64), allocatable :: fug_save(:)
        integer, allocatable :: act_save(:)
        real(real64) :: t_save
        fug_save = input%fugacity
        act_save = input%is_active
        t_save = input%temp_K
        input%is_active = 0
        input%is_active(1) = 1
        input%fugacity(1) = f_atm
        input%temp_K = temp_K
        call ConvertFugacity()
        f = input%fugacity(1)
        input%fugacity = fug_save
        input%is_active = act_save
        input%temp_K = t_save
    end function ref_convert_fugacity

    !---------------------------------------------------------------------------
    ! File-driven path: run the reference's own front end (main.f90:16-27) on a .maniac input,
    ! a LAMMPS .data topology and a .inc parameter file, up to and including
    ! PrepareSimulationParameters.  stage = 1 stops after ReadInput (input-file tests).
    ! The reference allocates its state once and never frees it, so this may be called ONCE per
    ! process (tests run it in a subprocess); a fatal input error ends the process with `stop`.
    !---------------------------------------------------------------------------
    function ref_load_files(maniac, data, inc, outdir, stage) bind(C, name="ref_load_files") result(rc)
        character(kind=c_char), intent(in) :: maniac(*), data(*), inc(*), outdir(*)
        integer(c_int), value :: stage
        integer(c_int) :: rc
        rc = 0
        if (is_setup) then
            rc = 1
            return
        end if
        maniac_file = cstr(maniac)
        data_file = cstr(data)
        inc_file = cstr(inc)
        output_path = cstr(outdir)
        res_file = pending_res_file
        call InitOutput()
        call ReadInput()
        if (stage > 1) then
            call ReadSystemData()
            call ReadParameters()
            call PrepareSimulationParameters()
        end if
        energy%ewald_self = zero
        is_setup = .true.
    contains
        function cstr(c) result(f)
            character(kind=c_char), intent(in) :: c(*)
            character(len=200) :: f
            integer :: i
            f = ''
            do i = 1, 200
                if (c(i) == c_null_char) exit
                f(i:i) = c(i)
            end do
        end function cstr
    end function ref_load_files


    !---------------------------------------------------------------------------
    ! Reservoir topology (-r of the reference's command line, cli_utils.f90:60-63): call before
    ! ref_load_files; an empty string means "no reservoir".
    !---------------------------------------------------------------------------
    subroutine ref_set_reservoir_file(path) bind(C, name="ref_set_reservoir_file")
        character(kind=c_char), intent(in) :: path(*)
        integer :: i
        pending_res_file = ''
        do i = 1, 200
            if (path(i) == c_null_char) exit
            pending_res_file(i:i) = path(i)
        end do
    end subroutine ref_set_reservoir_file

    !---------------------------------------------------------------------------
    ! The rest of program MANIAC (main.f90:26-33) after ref_load_files(stage = 2): PrecomputeTable,
    ! ComputeSystemEnergy, MonteCarloLoop, FinalReport -- the reference's own loop, writers and log.
    ! init_amplitude /= 0 sets A(k) <- S(k) first (the initialisation the reference omits, SURVEY
    ! F2; without it the per-move reciprocal energies are computed from heap garbage).  seed > 0
    ! re-seeds the intrinsic generator with the reference's own seed_rng (random_utils.f90:35-56);
    ! the reference itself skips that call whenever the input gives a seed (input_parser.f90:597).
    !---------------------------------------------------------------------------
    function ref_run_mc(seed, init_amplitude) bind(C, name="ref_run_mc") result(rc)
        integer(c_int), value :: seed, init_amplitude
        integer(c_int) :: rc
        rc = 1
        if (.not. is_setup) return
        call PrecomputeTable()
        energy%ewald_self = zero
        call ComputeSystemEnergy(primary)
        if (init_amplitude /= 0) call ref_init_amplitude(1_c_int)
        if (seed > 0) call seed_rng(int(seed))
        call MonteCarloLoop()
        call FinalReport()
        rc = 0
    end function ref_run_mc

    ! what ReadInput / ReadSystemData produced: sizes, then per-residue tables
    subroutine ref_get_sizes(n_res, max_atom, n_atom_types) bind(C, name="ref_get_sizes")
        integer(c_int), intent(out) :: n_res, max_atom, n_atom_types
        n_res = nb%type_residue
        max_atom = nb%max_atom_in_residue
        n_atom_types = primary%num_atomtypes
    end subroutine ref_get_sizes

    subroutine ref_get_residues(atoms_in_res, types_per_res, is_active, fugacity, n_mol) bind(C, name="ref_get_residues")
        integer(c_int), intent(out) :: atoms_in_res(nb%type_residue), types_per_res(nb%type_residue)
        integer(c_int), intent(out) :: is_active(nb%type_residue), n_mol(nb%type_residue)
        real(c_double), intent(out) :: fugacity(nb%type_residue)
        atoms_in_res = nb%atom_in_residue
        types_per_res = nb%types_per_residue
        is_active = input%is_active
        fugacity = input%fugacity
        if (allocated(primary%num_residues)) then
            n_mol = primary%num_residues
        else
            n_mol = 0
        end if
    end subroutine ref_get_residues

    ! site template of one residue type (1-based t): atom types and charges per site
    subroutine ref_get_template(t, atom_types, charges) bind(C, name="ref_get_template")
        integer(c_int), value :: t
        integer(c_int), intent(out) :: atom_types(nb%max_atom_in_residue)
        real(c_double), intent(out) :: charges(nb%max_atom_in_residue)
        atom_types = primary%atom_types(t, :)
        charges = primary%atom_charges(t, :)
    end subroutine ref_get_template


    ! bonded tables DetectBondPerResidue & co. built (data_parser.f90:320-550): kind 1 bonds (3 columns),
    ! 2 angles (4), 3 dihedrals (5), 4 impropers (5); table(c, k) for k <= n; also the "<kind> types" header counts
    subroutine ref_get_bonded(kind, t, n, table, n_types) bind(C, name="ref_get_bonded")
        integer(c_int), value :: kind, t
        integer(c_int), intent(out) :: n, table(5, 64), n_types
        integer :: k
        table = 0
        select case (kind)
        case (1)
            n = nb%bonds_per_residue(t); n_types = primary%num_bondtypes
            do k = 1, min(n, 64)
                table(1:3, k) = res%bond_type_2d(t, k, 1:3)
            end do
        case (2)
            n = nb%angles_per_residue(t); n_types = primary%num_angletypes
            do k = 1, min(n, 64)
                table(1:4, k) = res%angle_type_2d(t, k, 1:4)
            end do
        case (3)
            n = nb%dihedrals_per_residue(t); n_types = primary%num_dihedraltypes
            do k = 1, min(n, 64)
                table(1:5, k) = res%dihedral_type_2d(t, k, 1:5)
            end do
        case default
            n = nb%impropers_per_residue(t); n_types = primary%num_impropertypes
            do k = 1, min(n, 64)
                table(1:5, k) = res%improper_type_2d(t, k, 1:5)
            end do
        end select
    end subroutine ref_get_bonded

    ! epsilon / sigma of a site pair (1-based), after ReadParameters + ApplyLorentzBerthelot
    subroutine ref_get_coeff(t1, a1, t2, a2, eps, sig) bind(C, name="ref_get_coeff")
        integer(c_int), value :: t1, a1, t2, a2
        real(c_double), intent(out) :: eps, sig
        eps = coeff%epsilon(t1, t2, a1, a2)
        sig = coeff%sigma(t1, t2, a1, a2)
    end subroutine ref_get_coeff

    ! scalars of the .maniac input after parsing and rescaling
    subroutine ref_get_input(out) bind(C, name="ref_get_input")
        real(c_double), intent(out) :: out(12)
        out(1) = real(input%nb_block, real64)
        out(2) = real(input%nb_step, real64)
        out(3) = input%temp_K
        out(4) = input%ewald_tolerance
        out(5) = input%real_space_cutoff
        out(6) = input%translation_step
        out(7) = input%rotation_step_angle
        out(8) = merge(1.0_real64, 0.0_real64, input%recalibrate_moves)
        out(9) = proba%translation
        out(10) = proba%rotation
        out(11) = proba%insertion_deletion
        out(12) = proba%swap
    end subroutine ref_get_input

    subroutine ref_get_box_matrix(m, lo, hi) bind(C, name="ref_get_box_matrix")
        real(c_double), intent(out) :: m(9), lo(3), hi(3)
        m = reshape(primary%matrix, [9])
        lo = primary%bounds(:, 1)
        hi = primary%bounds(:, 2)
    end subroutine ref_get_box_matrix

    ! Compile-time constants of the reference (constants.f90:7-20)
    subroutine ref_constants(out) bind(C, name="ref_constants")
        real(c_double), intent(out) :: out(8)
        out(1) = PI
        out(2) = TWOPI
        out(3) = SQRTPI
        out(4) = EPS0_INV_eVA
        out(5) = KB_eVK
        out(6) = KB_kcalmol
        out(7) = error
        out(8) = real(NB_MAX_MOLECULE, real64)
    end subroutine ref_constants

end module ref_shim
