!===============================================================================
! mc_chain -- ONE Markov chain driven exactly like the reference's MonteCarloLoop, with the engine
! behind the reference-named seams of module maniac_gpu (the B = 1 drop-in of INTEGRATION.md).
!
! The control flow, the bookkeeping and -- deliberately -- the ORDER in which random numbers are drawn
! follow the reference statement by statement, so that with the same seed (seed_rng rule,
! src/random_utils.f90:35-56) and the same Fortran runtime the chain visits the same states as the
! reference and writes the same files:
!   MonteCarloLoop                    src/monte_carlo.f90:26-88
!   PickRandomResidueType / PickRandomMoleculeIndex   src/monte_carlo_utils.f90:136-182
!   Translation / RandomTranslation   src/translation.f90:36-112
!   Rotation / ApplyRandomRotation / ChooseRotationAngle   src/rotation.f90:34-75,
!                                     src/monte_carlo_utils.f90:30-92
!   CreateMolecule / InsertAndOrientMolecule / Accept / Reject   src/create_molecule.f90:41-207
!   DeleteMolecule / RemoveMolecule / Accept / Reject            src/delete_molecule.f90:41-199
!   ComputeOldEnergy / ComputeNewEnergy / AcceptMove             src/monte_carlo_utils.f90:275-422
!   mc_acceptance_probability         src/monte_carlo_utils.f90:184-226
!   AdjustMoveStepSizes               src/monte_carlo_utils.f90:99-130 (as written)
!   ApplyPBC                          src/geometry_utils.f90:167-213
! Differences from the reference, both deliberate (SURVEY F2 / F3): A(k) is initialised to S(k) by
! ComputeSystemEnergy, and a deletion's new reciprocal energy removes the deleted molecule.
! mchain_set_as_written(1) selects the reference's deletion exactly as written instead (F3): DeleteMolecule
! compacts first (RemoveMolecule: slot m <- last slot, delete_molecule.f90:99-116) and ComputeNewEnergy then calls
! ComputeRecipEnergySingleMol with is_creation = deletion_flag (monte_carlo_utils.f90:308), so A(k) GAINS the
! terms of the molecule now sitting in slot m and keeps them when the move is accepted.  That mode lives in
! this host loop only: it is composed from the engine's neutral primitives (a creation-kind reciprocal energy
! of the swapped-in molecule's sites; on acceptance mgpu_replica_replace_molecule, mgpu_replica_set_num_molecules
! and mgpu_structure_factor_add) -- no kernel knows about it.  Intended physics stays the default.
! Energies are evaluated against the engine's resident state; nothing is saved or restored on
! rejection (the reference's Save/RestoreSingleMolFourier have no counterpart).
!===============================================================================
module mc_chain

    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: real64
    use maniac_gpu
    use maniac_output

    implicit none

    private
    public :: mchain_reset, mchain_set_box, mchain_set_residue, mchain_set_bonded, mchain_set_tables, &
              mchain_set_moves, mchain_set_reservoir_box, mchain_set_reservoir_residue, mchain_run, &
              mchain_get_energy, mchain_get_counters, mchain_get_counts, mchain_get_molecule, mchain_get_steps, &
              mchain_set_mode, mchain_set_as_written, mchain_set_log_header, mchain_write_log_header, &
              mchain_set_speculation, mchain_set_chain_windows, mchain_get_loop_seconds, mchain_get_times

    real(real64), parameter :: PI = 3.14159265358979323846_real64, TWOPI = 2.0_real64 * PI
    real(real64), parameter :: zero = 0.0_real64, one = 1.0_real64, half = 0.5_real64, three = 3.0_real64
    ! src/parameters.f90:8, :14-21
    integer, parameter :: NB_MAX_MOLECULE = 5000
    real(real64), parameter :: TARGET_ACCEPTANCE = 0.40d0, TOL_ACCEPTANCE = 0.05d0
    real(real64), parameter :: MIN_TRANSLATION_STEP = 1.0d-3, MAX_TRANSLATION_STEP = 3.0d0
    real(real64), parameter :: MIN_ROTATION_ANGLE = 1.0d-3, MAX_ROTATION_ANGLE = 0.78d0
    real(real64), parameter :: PROB_CREATE_DELETE = 0.5d0
    integer, parameter :: MIN_TRIALS_FOR_RECALIBRATION = 500
    integer, parameter :: TYPE_CREATION = 1, TYPE_DELETION = 2, TYPE_TRANSLATION = 3, TYPE_ROTATION = 4

    type(chain_block), save, target :: S
    integer, save :: status = 0          ! first engine error met (0 = none); the loop stops on it
    ! .true.: one batched engine call per move (mgpu_gcmc_trial_submit / wait with one candidate, resident-row
    ! commit) instead of one call per reference seam -- same energies, a third of the launches and waits
    logical, save :: fused = .true.
    ! .true.: the reference's deletion update as written (SURVEY F3); see the header
    logical, save :: as_written = .false.
    ! the messages the reference logs before the Monte Carlo loop (banner, input echo, data-file summary,
    ! Lorentz-Berthelot listing, Ewald parameters), one per line, prepared by the front end (io_maniac.log_header_lines)
    character(len=:), allocatable, save :: log_header
    ! Speculative window (fused mode): the next spec_k steps are drawn in the reference's random-number order ASSUMING
    ! every one of them is rejected -- the state then does not change, so all their trials are trials from the current
    ! state and are evaluated in ONE batched engine call.  The window is then walked in order with the saved acceptance
    ! draws; the first accepted step is applied, the generator is put back to its state right after that step's draws
    ! and the rest of the window is thrown away.  Visited states, random stream and files are those of the sequential
    ! loop; the engine round trips per step drop by up to 1 / acceptance.  spec_k = 1: the sequential loop.
    integer, save :: spec_k = 1
    integer, parameter :: STEP_NOOP = -1, STEP_ABORT = -2
    ! One launch per window (mgpu_chain_window): the engine evaluates the window's steps, applies the acceptance rule to
    ! them in order with the draws saved here and commits the first accepted step itself; a step whose draw is too close
    ! to its probability for two exp implementations to be sure to agree comes back UNDECIDED and is decided (and
    ! committed) by this loop.  chain_windows: wanted (default); chain_cap: the engine's window capacity, 0 = the engine
    ! cannot (triclinic box, large molecules) and the window goes through mgpu_gcmc_trial_submit / wait as before.
    logical, save :: chain_windows = .true.
    integer, save :: chain_cap = 0
    integer, parameter :: BY_HOST = -1, DEVICE_REJECTED = 0, DEVICE_ACCEPTED = 1
    ! time spent inside the Monte Carlo loop proper (no initial energy, no files), seconds
    real(real64), save :: loop_seconds = 0.0_real64
    ! ... in ComputeSystemEnergy before the loop, and in the output files (UpdateFiles, PrintStatus)
    real(real64), save :: init_seconds = 0.0_real64, file_seconds = 0.0_real64

    type :: proposal
        integer :: t = 0, m = 0, kind = STEP_NOOP, mtype = 0
        integer :: cand = 0, cand2 = 0                  ! rows of the batch (cand2: the second row of an as-written deletion)
        integer :: pick = 0                             ! reservoir molecule an insertion copies
        real(real64) :: com(3) = 0.0_real64, u = 0.0_real64
        real(real64), allocatable :: off(:, :)          ! (3, n1)
        integer, allocatable :: rng(:)                  ! generator state after the step's draws
    end type proposal
    ! the window's candidate list as handed to the engine (kept for the resident-row commit)
    integer(c_int), allocatable, save :: w_rep(:), w_t(:), w_m(:), w_kind(:), w_acc(:), w_link(:)
    real(c_double), allocatable, save :: w_u(:), w_pref(:)
    integer, save :: w_stride = 1

contains

    ! rand_uniform / rand_symmetric (src/random_utils.f90:13-31)
    function rand_uniform() result(v)
        real(real64) :: v
        call random_number(v)
    end function rand_uniform

    subroutine seed_rng(seed)
        integer, intent(in) :: seed
        integer :: n, i
        integer, allocatable :: put(:)
        call random_seed(size=n)
        allocate(put(n))
        put = seed + 37 * [(i - 1, i = 1, n)]
        call random_seed(put=put)
    end subroutine seed_rng

    subroutine note(stat)
        integer, intent(in) :: stat
        if (stat /= 0 .and. status == 0) status = stat
    end subroutine note

    !---------------------------------------------------------------------------
    ! Building the state (called from the host plumbing, in this order: reset, box, residues,
    ! bonded tables, tables, moves, reservoir)
    !---------------------------------------------------------------------------
    subroutine mchain_reset(engine, n_res, n_atom_types, temperature) bind(C, name="mchain_reset")
        type(c_ptr), value :: engine
        integer(c_int), value :: n_res, n_atom_types
        real(c_double), value :: temperature
        call blank_chain(S)
        S%engine = engine
        S%n_res = n_res
        S%n_atom_types = n_atom_types
        S%temperature = temperature
        allocate(S%res(n_res), S%rsv(n_res), S%masses(n_atom_types))
        S%masses = zero
        status = 0
    end subroutine mchain_reset

    subroutine blank_chain(c)
        type(chain_block), intent(out) :: c            ! intent(out): every component back to its default
        c%engine = c_null_ptr
    end subroutine blank_chain

    subroutine fill_box(b, matrix, reciprocal, lo, hi, tilt, is_triclinic, box_type, volume)
        type(box_block), intent(out) :: b
        real(c_double), intent(in) :: matrix(3, 3), reciprocal(3, 3), lo(3), hi(3), tilt(3), volume
        integer(c_int), intent(in) :: is_triclinic, box_type
        b%matrix = matrix
        b%reciprocal = reciprocal
        b%lo = lo
        b%hi = hi
        b%tilt = tilt
        b%is_triclinic = is_triclinic /= 0
        b%box_type = box_type
        b%volume = volume
        b%num_atoms = 0
    end subroutine fill_box

    ! matrix / reciprocal are the Fortran arrays box%matrix / box%reciprocal (column-major)
    subroutine mchain_set_box(matrix, reciprocal, lo, hi, tilt, is_triclinic, box_type, volume) &
            bind(C, name="mchain_set_box")
        real(c_double), intent(in) :: matrix(3, 3), reciprocal(3, 3), lo(3), hi(3), tilt(3)
        integer(c_int), value :: is_triclinic, box_type
        real(c_double), value :: volume
        call fill_box(S%box, matrix, reciprocal, lo, hi, tilt, is_triclinic, box_type, volume)
    end subroutine mchain_set_box

    subroutine mchain_set_reservoir_box(matrix, reciprocal, lo, hi, tilt, is_triclinic, box_type, volume, &
                                        any_bonded) bind(C, name="mchain_set_reservoir_box")
        real(c_double), intent(in) :: matrix(3, 3), reciprocal(3, 3), lo(3), hi(3), tilt(3)
        integer(c_int), value :: is_triclinic, box_type
        real(c_double), value :: volume
        integer(c_int), intent(in) :: any_bonded(4)
        call fill_box(S%rbox, matrix, reciprocal, lo, hi, tilt, is_triclinic, box_type, volume)
        S%has_reservoir = .true.
        S%any_bonded = any_bonded /= 0
    end subroutine mchain_set_reservoir_box

    subroutine fill_residue(r, n1, cap, count, com, off)
        type(residue_block), intent(inout) :: r
        integer, intent(in) :: n1, cap, count
        real(c_double), intent(in) :: com(3, *), off(3, n1, *)
        integer :: m
        r%n1 = n1
        r%cap = max(cap, count, 1)
        r%count = count
        if (allocated(r%com)) deallocate(r%com, r%off)
        allocate(r%com(3, r%cap), r%off(3, n1, r%cap))
        r%com = zero
        r%off = zero
        do m = 1, count
            r%com(:, m) = com(:, m)
            r%off(:, :, m) = off(:, :, m)
        end do
    end subroutine fill_residue

    subroutine mchain_set_residue(t, name, n1, active, cap, count, atom_type, charge, com, off, fugacity) &
            bind(C, name="mchain_set_residue")
        integer(c_int), value :: t, n1, active, cap, count
        character(kind=c_char), intent(in) :: name(*)
        integer(c_int), intent(in) :: atom_type(n1)
        real(c_double), intent(in) :: charge(n1), com(3, *), off(3, n1, *)
        real(c_double), value :: fugacity
        integer :: i
        associate (r => S%res(t))
            r%name = ''
            do i = 1, 10
                if (name(i) == c_null_char) exit
                r%name(i:i) = name(i)
            end do
            r%active = active
            r%atom_type = atom_type
            r%charge = charge
            r%fugacity = fugacity
            call fill_residue(r, int(n1), int(cap), int(count), com, off)
            S%box%num_atoms = S%box%num_atoms + count * n1
        end associate
    end subroutine mchain_set_residue

    subroutine mchain_set_reservoir_residue(t, n1, cap, count, com, off) bind(C, name="mchain_set_reservoir_residue")
        integer(c_int), value :: t, n1, cap, count
        real(c_double), intent(in) :: com(3, *), off(3, n1, *)
        call fill_residue(S%rsv(t), int(n1), int(cap), int(count), com, off)
        S%rbox%num_atoms = S%rbox%num_atoms + count * n1
    end subroutine mchain_set_reservoir_residue

    ! kind 1 bonds, 2 angles, 3 dihedrals, 4 impropers; table(1, k) = type, table(2:, k) = local atom indices
    subroutine mchain_set_bonded(t, kind, n, table) bind(C, name="mchain_set_bonded")
        integer(c_int), value :: t, kind, n
        integer(c_int), intent(in) :: table(5, *)
        integer, allocatable :: grown(:, :, :)
        integer :: k
        associate (r => S%res(t))
            if (.not. allocated(r%bonded)) then
                allocate(r%bonded(5, max(1, n), 4))
                r%bonded = 0
            else if (size(r%bonded, 2) < n) then
                allocate(grown(5, n, 4))
                grown = 0
                grown(:, 1:size(r%bonded, 2), :) = r%bonded
                call move_alloc(grown, r%bonded)
            end if
            r%n_bonded(kind) = n
            do k = 1, n
                r%bonded(:, k, kind) = table(:, k)
            end do
        end associate
    end subroutine mchain_set_bonded

    subroutine mchain_set_tables(masses, n_bonded_types) bind(C, name="mchain_set_tables")
        real(c_double), intent(in) :: masses(*)
        integer(c_int), intent(in) :: n_bonded_types(4)
        S%masses = masses(1:S%n_atom_types)
        S%n_bonded_types = n_bonded_types
    end subroutine mchain_set_tables

    subroutine mchain_set_moves(translation_step, rotation_step, p_translation, p_rotation, recalibrate) &
            bind(C, name="mchain_set_moves")
        real(c_double), value :: translation_step, rotation_step, p_translation, p_rotation
        integer(c_int), value :: recalibrate
        S%translation_step = translation_step
        S%rotation_step = rotation_step
        S%p_translation = p_translation
        S%p_rotation = p_rotation
        S%recalibrate = recalibrate /= 0
    end subroutine mchain_set_moves

    !---------------------------------------------------------------------------
    ! Geometry helpers
    !---------------------------------------------------------------------------
    subroutine apply_pbc(pos, box)
        real(real64), intent(inout) :: pos(3)
        type(box_block), intent(in) :: box
        real(real64) :: frac(3)
        integer :: d
        if (.not. box%is_triclinic) then
            do d = 1, 3
                pos(d) = box%lo(d) + modulo(pos(d) - box%lo(d), box%matrix(d, d))
            end do
        else
            frac = matmul(box%reciprocal, pos - box%lo)
            do d = 1, 3
                frac(d) = modulo(frac(d), one)
            end do
            pos = box%lo + matmul(box%matrix, frac)
        end if
    end subroutine apply_pbc

    ! RotationMatrix (src/helper_utils.f90:39-77)
    function axis_rotation(axis, theta) result(r)
        integer, intent(in) :: axis
        real(real64), intent(in) :: theta
        real(real64) :: r(3, 3), c, sn
        integer :: p, q
        c = cos(theta)
        sn = sin(theta)
        r = zero
        r(1, 1) = one; r(2, 2) = one; r(3, 3) = one
        ! the plane (p, q) the axis leaves invariant: x -> (2,3), y -> (3,1), z -> (1,2)
        p = mod(axis, 3) + 1
        q = mod(axis + 1, 3) + 1
        if (axis >= 1 .and. axis <= 3) then
            r(p, p) = c; r(p, q) = -sn
            r(q, p) = sn; r(q, q) = c
        end if
    end function axis_rotation

    ! ApplyRandomRotation: the angle is drawn before the axis
    subroutine random_rotation(t, m, full)
        integer, intent(in) :: t, m
        logical, intent(in) :: full
        real(real64) :: theta, rot(3, 3)
        integer :: axis, n1
        n1 = S%res(t)%n1
        if (n1 == 1) return
        if (full) then
            theta = rand_uniform() * TWOPI
        else
            theta = (rand_uniform() - half) * S%rotation_step
        end if
        axis = int(rand_uniform() * three) + 1
        rot = axis_rotation(axis, theta)
        S%res(t)%off(:, 1:n1, m) = matmul(rot, S%res(t)%off(:, 1:n1, m))
    end subroutine random_rotation

    !---------------------------------------------------------------------------
    ! Energies of one molecule: e = non_coulomb, coulomb, recip, self, intra, total
    !---------------------------------------------------------------------------
    subroutine molecule_energy(t, m, e, creation, deletion, is_new)
        integer, intent(in) :: t, m
        real(real64), intent(out) :: e(6)
        logical, intent(in) :: creation, deletion, is_new
        integer :: n1, stat, exclude
        n1 = S%res(t)%n1
        e = zero
        if (creation .and. .not. is_new) then
            e(IE_RECIP) = S%energy(IE_RECIP)                              ! ComputeOldEnergy, creation branch
        else if (deletion .and. is_new) then
            ! ComputeNewEnergy, deletion branch: only the reciprocal energy of the system without the
            ! molecule (the engine still holds it in slot m)
            call ComputeRecipEnergySingleMol(S%engine, t, m, S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, &
                                             e(IE_RECIP), is_deletion=.true., stat=stat)
            call note(stat)
        else
            if (creation) then
                call ComputeRecipEnergySingleMol(S%engine, t, m, S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, &
                                                 e(IE_RECIP), is_creation=.true., stat=stat)
            else if (deletion) then
                e(IE_RECIP) = S%energy(IE_RECIP)                          ! ComputeOldEnergy, deletion branch
                stat = 0
            else
                call ComputeRecipEnergySingleMol(S%engine, t, m, S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, &
                                                 e(IE_RECIP), stat=stat)
            end if
            call note(stat)
            exclude = m
            if (creation) exclude = 0                                   ! the engine does not hold the molecule yet
            call ComputePairInteractionEnergy_singlemol(S%engine, t, exclude, S%res(t)%com(:, m), &
                                                        S%res(t)%off(:, 1:n1, m), n1, e(IE_NONC), e(IE_COUL), stat=stat)
            call note(stat)
            if (creation .or. deletion) then
                call ComputeEwaldSelfInteractionSingleMol(S%engine, t, e(IE_SELF), stat=stat)
                call note(stat)
                call ComputeIntraResidueRealCoulombEnergySingleMol(S%engine, t, m, S%res(t)%com(:, m), &
                                                                   S%res(t)%off(:, 1:n1, m), n1, e(IE_INTRA), stat=stat)
                call note(stat)
            end if
        end if
        if (creation .or. deletion) then
            e(IE_TOTAL) = e(IE_NONC) + e(IE_COUL) + e(IE_RECIP) + e(IE_SELF) + e(IE_INTRA)
        else
            e(IE_TOTAL) = e(IE_NONC) + e(IE_COUL) + e(IE_RECIP)
        end if
    end subroutine molecule_energy

    ! seams /= 0: call the reference-named seams one by one (the literal integration of INTEGRATION.md section 3)
    subroutine mchain_set_mode(seams) bind(C, name="mchain_set_mode")
        integer(c_int), value :: seams
        fused = seams == 0
    end subroutine mchain_set_mode

    ! text = the header messages separated by line feeds (n bytes; n = 0 clears it)
    subroutine mchain_set_log_header(text, n) bind(C, name="mchain_set_log_header")
        integer(c_int), value :: n
        character(kind=c_char), intent(in) :: text(*)
        integer :: i
        if (allocated(log_header)) deallocate(log_header)
        if (n <= 0) return
        allocate(character(len=n) :: log_header)
        do i = 1, n
            log_header(i:i) = text(i)
        end do
    end subroutine mchain_set_log_header

    ! every header message through the reference's list-directed write (LogMessage, output_utils.f90:30-36)
    subroutine write_log_header(ch)
        type(chain_block), intent(in) :: ch
        integer :: a, b, n
        if (.not. allocated(log_header)) return
        n = len(log_header)
        a = 1
        do while (a <= n + 1)
            b = index(log_header(min(a, n):n), achar(10))
            if (a > n) then
                call log_line(ch, '')
                exit
            end if
            if (b == 0) then
                call log_line(ch, log_header(a:n))
                exit
            end if
            call log_line(ch, log_header(a:a + b - 2))
            a = a + b
        end do
    end subroutine write_log_header

    ! Test hook (no engine needed): write the header alone to `path`
    subroutine mchain_write_log_header(path) bind(C, name="mchain_write_log_header")
        character(kind=c_char), intent(in) :: path(*)
        character(len=512) :: p
        integer :: i
        type(chain_block) :: tmp
        p = ''
        do i = 1, len(p)
            if (path(i) == c_null_char) exit
            p(i:i) = path(i)
        end do
        tmp%log_unit = 37
        open(unit=tmp%log_unit, file=trim(p), status='replace')
        call write_log_header(tmp)
        close(tmp%log_unit)
    end subroutine mchain_write_log_header

    ! k > 1: speculative windows of k steps in the batched (non-seams) mode; k <= 1: one step at a time
    subroutine mchain_set_speculation(k) bind(C, name="mchain_set_speculation")
        integer(c_int), value :: k
        spec_k = max(1, min(int(k), 64))
    end subroutine mchain_set_speculation

    ! on /= 0 (default): windows go to the engine's one-launch path where it applies
    subroutine mchain_set_chain_windows(on) bind(C, name="mchain_set_chain_windows")
        integer(c_int), value :: on
        chain_windows = on /= 0
    end subroutine mchain_set_chain_windows

    function mchain_get_loop_seconds() bind(C, name="mchain_get_loop_seconds") result(sec)
        real(c_double) :: sec
        sec = loop_seconds
    end function mchain_get_loop_seconds

    ! seconds of the last mchain_run: initial energy, Monte Carlo steps, output files
    subroutine mchain_get_times(t) bind(C, name="mchain_get_times")
        real(c_double), intent(out) :: t(3)
        t = [init_seconds, loop_seconds, file_seconds]
    end subroutine mchain_get_times

    subroutine mchain_set_as_written(on) bind(C, name="mchain_set_as_written")
        integer(c_int), value :: on
        as_written = on /= 0
    end subroutine mchain_set_as_written

    !---------------------------------------------------------------------------
    ! Both energy states of a move from ONE engine call (fused mode): the candidate's sites are the
    ! molecule's current com + offsets in the host state; `old` / `new` are filled exactly as
    ! ComputeOldEnergy / ComputeNewEnergy fill them for that move type.
    !---------------------------------------------------------------------------
    subroutine fused_energies(t, m, kind, old, new)
        integer, intent(in) :: t, m, kind
        real(real64), intent(out) :: old(6), new(6)
        integer :: n1
        integer(c_int) :: rep(1), tt(1), mm(1), kk(1), rc
        real(real64) :: o(5), w(5)
        real(real64), allocatable :: sites(:, :)
        n1 = S%res(t)%n1
        allocate(sites(3, n1))
        call MoleculeSites(S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, sites)
        rep = 0; tt = t - 1; mm = m - 1; kk = int(kind, c_int)
        if (kind == MGPU_CREATION) mm = -1
        rc = mgpu_gcmc_trial_submit(S%engine, 0_c_int, 1_c_int, rep, tt, mm, kk, sites, int(n1, c_int))
        if (rc == MGPU_OK) rc = mgpu_gcmc_trial_wait(S%engine, 0_c_int, o, w)
        call note(int(rc))
        old = zero
        new = zero
        select case (kind)
        case (MGPU_CREATION)
            old(IE_RECIP) = S%energy(IE_RECIP)
            new(1:5) = w
        case (MGPU_DELETION)
            old(1:5) = o
            old(IE_RECIP) = S%energy(IE_RECIP)
            new(IE_RECIP) = w(IE_RECIP)
        case default
            old(1:3) = o(1:3)
            new(1:3) = w(1:3)
        end select
        if (kind == MGPU_MOVE) then
            old(IE_TOTAL) = old(IE_NONC) + old(IE_COUL) + old(IE_RECIP)
            new(IE_TOTAL) = new(IE_NONC) + new(IE_COUL) + new(IE_RECIP)
        else
            old(IE_TOTAL) = old(IE_NONC) + old(IE_COUL) + old(IE_RECIP) + old(IE_SELF) + old(IE_INTRA)
            new(IE_TOTAL) = new(IE_NONC) + new(IE_COUL) + new(IE_RECIP) + new(IE_SELF) + new(IE_INTRA)
        end if
    end subroutine fused_energies

    ! apply the move the lane has just evaluated (fused mode): the rows are still on the device
    subroutine fused_commit(t, m, kind)
        integer, intent(in) :: t, m, kind
        integer(c_int) :: rep(1), tt(1), mm(1), kk(1), acc(1), rc
        rep = 0; tt = t - 1; mm = m - 1; kk = int(kind, c_int); acc = 1
        if (kind == MGPU_CREATION) mm = -1
        rc = mgpu_commit_submit(S%engine, 0_c_int, 1_c_int, rep, tt, mm, kk, c_null_ptr, int(S%res(t)%n1, c_int), acc)
        call note(int(rc))
    end subroutine fused_commit

    function acceptance_probability(old, new, t, move_type) result(p)
        real(real64), intent(in) :: old(6), new(6)
        integer, intent(in) :: t, move_type
        real(real64) :: p, n, v, phi, temp, delta_e
        n = real(S%res(t)%count)
        v = S%box%volume
        phi = S%res(t)%fugacity
        temp = S%temperature
        delta_e = new(IE_TOTAL) - old(IE_TOTAL)
        select case (move_type)
        case (TYPE_CREATION)
            p = min(one, (phi * v / n) * exp(-delta_e / temp))
        case (TYPE_DELETION)
            p = min(one, ((n + one) / (phi * v)) * exp(-delta_e / temp))
        case default
            p = min(one, exp(-delta_e / temp))
        end select
    end function acceptance_probability

    ! The factor in front of exp(-delta_e / T) in mc_acceptance_probability, formed exactly as acceptance_probability forms
    ! it; n_after = the type's molecule count as the move drivers have it when they call the rule (after the insertion /
    ! after the removal)
    function acceptance_prefactor(t, move_type, n_after) result(f)
        integer, intent(in) :: t, move_type, n_after
        real(real64) :: f, n, v, phi
        n = real(n_after)
        v = S%box%volume
        phi = S%res(t)%fugacity
        select case (move_type)
        case (TYPE_CREATION)
            f = (phi * v / n)
        case (TYPE_DELETION)
            f = ((n + one) / (phi * v))
        case default
            f = one
        end select
    end function acceptance_prefactor

    ! AcceptMove: running energies, then the engine applies the move
    subroutine accept_move(t, m, old, new, which)
        integer, intent(in) :: t, m, which
        real(real64), intent(in) :: old(6), new(6)
        integer :: stat, n1
        S%energy(IE_RECIP) = S%energy(IE_RECIP) + new(IE_RECIP) - old(IE_RECIP)
        S%energy(IE_NONC) = S%energy(IE_NONC) + new(IE_NONC) - old(IE_NONC)
        S%energy(IE_COUL) = S%energy(IE_COUL) + new(IE_COUL) - old(IE_COUL)
        S%energy(IE_TOTAL) = S%energy(IE_TOTAL) + new(IE_TOTAL) - old(IE_TOTAL)
        S%counter(which) = S%counter(which) + 1
        n1 = S%res(t)%n1
        if (fused) then
            call fused_commit(t, m, MGPU_MOVE)
        else
            call GpuAcceptMove(S%engine, t, m, MGPU_MOVE, S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, stat=stat)
            call note(stat)
        end if
    end subroutine accept_move

    subroutine translation(t, m)
        integer, intent(in) :: t, m
        real(real64) :: com_old(3), old(6), new(6), trial(3), p
        if (m == 0) return
        S%counter(C_TRIAL_T) = S%counter(C_TRIAL_T) + 1
        com_old = S%res(t)%com(:, m)
        if (.not. fused) call molecule_energy(t, m, old, .false., .false., .false.)
        call random_number(trial)
        trial = (trial - half) * S%translation_step
        S%res(t)%com(:, m) = S%res(t)%com(:, m) + trial
        call apply_pbc(S%res(t)%com(:, m), S%box)
        if (fused) then
            call fused_energies(t, m, MGPU_MOVE, old, new)
        else
            call molecule_energy(t, m, new, .false., .false., .true.)
        end if
        p = acceptance_probability(old, new, t, TYPE_TRANSLATION)
        if (rand_uniform() <= p) then
            call accept_move(t, m, old, new, C_T)
        else
            S%res(t)%com(:, m) = com_old
        end if
    end subroutine translation

    subroutine rotation(t, m)
        integer, intent(in) :: t, m
        real(real64) :: old(6), new(6), p
        real(real64), allocatable :: off_old(:, :)
        if (S%res(t)%n1 == 1 .or. m == 0) return
        S%counter(C_TRIAL_R) = S%counter(C_TRIAL_R) + 1
        off_old = S%res(t)%off(:, :, m)
        if (.not. fused) call molecule_energy(t, m, old, .false., .false., .false.)
        call random_rotation(t, m, .false.)
        if (fused) then
            call fused_energies(t, m, MGPU_MOVE, old, new)
        else
            call molecule_energy(t, m, new, .false., .false., .true.)
        end if
        p = acceptance_probability(old, new, t, TYPE_ROTATION)
        if (rand_uniform() <= p) then
            call accept_move(t, m, old, new, C_R)
        else
            S%res(t)%off(:, :, m) = off_old
        end if
    end subroutine rotation

    subroutine create_molecule(t, m)
        integer, intent(in) :: t, m
        real(real64) :: old(6), new(6), p, trial(3), u
        integer :: n1, pick, last, stat
        n1 = S%res(t)%n1
        if (m > NB_MAX_MOLECULE .or. m > S%res(t)%cap) then            ! CheckMoleculeIndex aborts the reference here
            call note(4)
            return
        end if
        S%counter(C_TRIAL_C) = S%counter(C_TRIAL_C) + 1
        if (.not. fused) call molecule_energy(t, m, old, .true., .false., .false.)
        S%res(t)%count = S%res(t)%count + 1
        S%box%num_atoms = S%box%num_atoms + n1
        ! InsertAndOrientMolecule
        call random_number(trial)
        S%res(t)%com(:, m) = S%box%lo + matmul(S%box%matrix, trial)
        pick = 0
        if (S%has_reservoir) then
            call random_number(u)
            pick = int(u * S%rsv(t)%count) + 1
            S%res(t)%off(:, 1:n1, m) = S%rsv(t)%off(:, 1:n1, pick)
        else
            S%res(t)%off(:, 1:n1, m) = S%res(t)%off(:, 1:n1, 1)
            call random_rotation(t, m, .true.)
        end if
        if (fused) then
            call fused_energies(t, m, MGPU_CREATION, old, new)
        else
            call molecule_energy(t, m, new, .true., .false., .true.)
        end if
        p = acceptance_probability(old, new, t, TYPE_CREATION)
        if (rand_uniform() <= p) then
            S%energy(IE_RECIP) = new(IE_RECIP)
            S%energy(IE_NONC) = S%energy(IE_NONC) + new(IE_NONC) - old(IE_NONC)
            S%energy(IE_COUL) = S%energy(IE_COUL) + new(IE_COUL) - old(IE_COUL)
            S%energy(IE_SELF) = S%energy(IE_SELF) + new(IE_SELF) - old(IE_SELF)
            S%energy(IE_INTRA) = S%energy(IE_INTRA) + new(IE_INTRA) - old(IE_INTRA)
            S%energy(IE_TOTAL) = S%energy(IE_TOTAL) + new(IE_TOTAL) - old(IE_TOTAL)
            S%counter(C_C) = S%counter(C_C) + 1
            if (fused) then
                call fused_commit(t, m, MGPU_CREATION)
            else
                call GpuAcceptMove(S%engine, t, m, MGPU_CREATION, S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, stat=stat)
                call note(stat)
            end if
            if (S%has_reservoir) then
                ! the copied molecule leaves the reservoir: its slot takes the reservoir's last molecule
                last = S%rsv(t)%count
                S%rsv(t)%com(:, pick) = S%rsv(t)%com(:, last)
                S%rsv(t)%off(:, 1:n1, pick) = S%rsv(t)%off(:, 1:n1, last)
                S%rsv(t)%count = S%rsv(t)%count - 1
                S%rbox%num_atoms = S%rbox%num_atoms - n1
            end if
        else
            S%box%num_atoms = S%box%num_atoms - n1
            S%res(t)%count = S%res(t)%count - 1
        end if
    end subroutine create_molecule

    subroutine delete_molecule(t, m)
        integer, intent(in) :: t, m
        real(real64) :: old(6), new(6), p, trial(3), com_old(3)
        real(real64), allocatable :: off_old(:, :), off_last(:, :), sites_last(:, :)
        integer :: n1, last, stat
        if (S%res(t)%count == 0) return
        n1 = S%res(t)%n1
        S%counter(C_TRIAL_D) = S%counter(C_TRIAL_D) + 1
        if (.not. fused) call molecule_energy(t, m, old, .false., .true., .false.)
        com_old = S%res(t)%com(:, m)
        off_old = S%res(t)%off(:, :, m)
        last = S%res(t)%count
        ! the engine evaluates the removal of the molecule that sits in slot m, so the new energy is
        ! taken before the host mirror is compacted (RemoveMolecule in the reference comes first)
        if (fused) then
            call fused_energies(t, m, MGPU_DELETION, old, new)
        else
            call molecule_energy(t, m, new, .false., .true., .true.)
        end if
        if (as_written) then
            ! ComputeNewEnergy as written (monte_carlo_utils.f90:301-309): after RemoveMolecule slot m holds the
            ! former last molecule, and the reciprocal update runs with is_creation = .true. on that slot
            new = zero
            call ComputeRecipEnergySingleMol(S%engine, t, last, S%res(t)%com(:, last), S%res(t)%off(:, 1:n1, last), n1, &
                                             new(IE_RECIP), is_creation=.true., stat=stat)
            call note(stat)
            new(IE_TOTAL) = new(IE_NONC) + new(IE_COUL) + new(IE_RECIP) + new(IE_SELF) + new(IE_INTRA)
        end if
        off_last = S%res(t)%off(:, :, last)
        S%res(t)%com(:, m) = S%res(t)%com(:, last)
        S%res(t)%off(:, :, m) = S%res(t)%off(:, :, last)
        S%res(t)%count = S%res(t)%count - 1
        S%box%num_atoms = S%box%num_atoms - n1
        p = acceptance_probability(old, new, t, TYPE_DELETION)
        if (rand_uniform() <= p) then
            S%energy(IE_RECIP) = new(IE_RECIP)
            S%energy(IE_NONC) = S%energy(IE_NONC) + new(IE_NONC) - old(IE_NONC)
            S%energy(IE_COUL) = S%energy(IE_COUL) + new(IE_COUL) - old(IE_COUL)
            S%energy(IE_SELF) = S%energy(IE_SELF) + new(IE_SELF) - old(IE_SELF)
            S%energy(IE_INTRA) = S%energy(IE_INTRA) + new(IE_INTRA) - old(IE_INTRA)
            S%energy(IE_TOTAL) = S%energy(IE_TOTAL) + new(IE_TOTAL) - old(IE_TOTAL)
            S%counter(C_D) = S%counter(C_D) + 1
            if (as_written) then
                ! coordinates and count as RemoveMolecule leaves them; A(k) keeps the swapped-in molecule's terms
                allocate(sites_last(3, n1))
                call MoleculeSites(S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, sites_last)   ! slot m now = former last
                stat = mgpu_replica_replace_molecule(S%engine, 0_c_int, int(t - 1, c_int), int(m - 1, c_int), int(last - 1, c_int))
                call note(stat)
                stat = mgpu_replica_set_num_molecules(S%engine, 0_c_int, int(t - 1, c_int), int(last - 1, c_int))
                call note(stat)
                stat = mgpu_structure_factor_add(S%engine, 0_c_int, int(t - 1, c_int), sites_last)
                call note(stat)
                deallocate(sites_last)
            else if (fused) then
                call fused_commit(t, m, MGPU_DELETION)
            else
                call GpuAcceptMove(S%engine, t, m, MGPU_DELETION, com_old, off_old(:, 1:n1), n1, stat=stat)
                call note(stat)
            end if
            if (S%has_reservoir) then
                ! the reservoir receives the geometry stored in the primary box's last slot, at a random place
                call random_number(trial)
                trial = trial - half
                associate (r => S%rsv(t))
                    if (r%count + 1 <= r%cap) then
                        r%com(:, r%count + 1) = trial(1) * S%rbox%matrix(:, 1) + trial(2) * S%rbox%matrix(:, 2) + &
                                                trial(3) * S%rbox%matrix(:, 3)
                        r%off(:, 1:n1, r%count + 1) = off_last(:, 1:n1)
                        r%count = r%count + 1
                        S%rbox%num_atoms = S%rbox%num_atoms + n1
                    else
                        call note(4)
                    end if
                end associate
            end if
        else
            S%res(t)%count = S%res(t)%count + 1
            S%box%num_atoms = S%box%num_atoms + n1
            S%res(t)%com(:, m) = com_old
            S%res(t)%off(:, :, m) = off_old
        end if
    end subroutine delete_molecule


    !---------------------------------------------------------------------------
    ! Speculative window.  propose_step draws one step's random numbers exactly as the loop body of MonteCarloLoop and
    ! the move drivers draw them (type, molecule, move, [insertion / deletion], displacement | angle then axis | position
    ! then pick / angle then axis, acceptance) and builds the trial geometry WITHOUT touching the chain's state.
    !---------------------------------------------------------------------------
    subroutine propose_step(p)
        type(proposal), intent(out) :: p
        integer :: t, m, n1, axis, nseed
        real(real64) :: draw, trial(3), theta, rot(3, 3), u
        t = pick_residue_type()
        m = pick_molecule_index(S%res(t)%count)
        draw = rand_uniform()
        n1 = S%res(t)%n1
        p%t = t
        p%m = m
        p%kind = STEP_NOOP
        if (draw <= S%p_translation) then
            if (m /= 0) then                                             ! Translation returns at once for an empty type
                p%mtype = TYPE_TRANSLATION
                p%kind = MGPU_MOVE
                call random_number(trial)
                trial = (trial - half) * S%translation_step
                p%com = S%res(t)%com(:, m) + trial
                call apply_pbc(p%com, S%box)
                p%off = S%res(t)%off(:, 1:n1, m)
                p%u = rand_uniform()
            end if
        else if (draw <= S%p_rotation + S%p_translation) then
            if (n1 /= 1 .and. m /= 0) then
                p%mtype = TYPE_ROTATION
                p%kind = MGPU_MOVE
                p%com = S%res(t)%com(:, m)
                theta = (rand_uniform() - half) * S%rotation_step        ! ApplyRandomRotation: angle, then axis
                axis = int(rand_uniform() * three) + 1
                rot = axis_rotation(axis, theta)
                p%off = matmul(rot, S%res(t)%off(:, 1:n1, m))
                p%u = rand_uniform()
            end if
        else
            if (rand_uniform() <= PROB_CREATE_DELETE) then
                m = S%res(t)%count + 1
                p%m = m
                p%mtype = TYPE_CREATION
                if (m > NB_MAX_MOLECULE .or. m > S%res(t)%cap) then     ! CheckMoleculeIndex aborts the reference here
                    p%kind = STEP_ABORT
                else
                    p%kind = MGPU_CREATION
                    call random_number(trial)
                    p%com = S%box%lo + matmul(S%box%matrix, trial)
                    if (S%has_reservoir) then
                        call random_number(u)
                        p%pick = int(u * S%rsv(t)%count) + 1
                        p%off = S%rsv(t)%off(:, 1:n1, p%pick)
                    else
                        p%off = S%res(t)%off(:, 1:n1, 1)
                        if (n1 /= 1) then
                            theta = rand_uniform() * TWOPI
                            axis = int(rand_uniform() * three) + 1
                            rot = axis_rotation(axis, theta)
                            p%off = matmul(rot, p%off)
                        end if
                    end if
                    p%u = rand_uniform()
                end if
            else
                if (S%res(t)%count /= 0) then                            ! DeleteMolecule returns for an empty type
                    p%mtype = TYPE_DELETION
                    p%kind = MGPU_DELETION
                    p%com = S%res(t)%com(:, m)
                    p%off = S%res(t)%off(:, 1:n1, m)
                    p%u = rand_uniform()
                end if
            end if
        end if
        call random_seed(size=nseed)
        allocate(p%rng(nseed))
        call random_seed(get=p%rng)
    end subroutine propose_step

    ! Walk one evaluated step: o / w = the engine's old / new rows of its candidate (w2: the second row of an as-written
    ! deletion).  Mirrors the part of Translation / Rotation / CreateMolecule / DeleteMolecule that follows the energy
    ! evaluation; `restore` = the generator must be put back to this step's state before anything else is drawn.
    ! verdict: BY_HOST -- the rule is applied here and an accepted step is committed through the engine (from the lane's
    ! resident rows, or with explicit sites after a one-launch window); DEVICE_REJECTED / DEVICE_ACCEPTED -- the engine has
    ! applied the rule (and the commit) already: only the chain's own bookkeeping follows.
    subroutine resolve_step(p, o, w, w2, n_cand, restore, accepted, verdict)
        type(proposal), intent(in) :: p
        real(real64), intent(in) :: o(5), w(5), w2(5)
        integer, intent(in) :: n_cand, verdict
        logical, intent(in) :: restore
        logical, intent(out) :: accepted
        real(real64) :: old(6), new(6), prob, trial(3)
        real(real64), allocatable :: off_last(:, :), sites_last(:, :)
        integer :: t, m, n1, last, stat
        t = p%t
        m = p%m
        n1 = S%res(t)%n1
        old = zero
        new = zero
        accepted = .false.
        select case (p%mtype)
        case (TYPE_TRANSLATION, TYPE_ROTATION)
            if (p%mtype == TYPE_TRANSLATION) then
                S%counter(C_TRIAL_T) = S%counter(C_TRIAL_T) + 1
            else
                S%counter(C_TRIAL_R) = S%counter(C_TRIAL_R) + 1
            end if
            old(1:3) = o(1:3)
            new(1:3) = w(1:3)
            old(IE_TOTAL) = old(IE_NONC) + old(IE_COUL) + old(IE_RECIP)
            new(IE_TOTAL) = new(IE_NONC) + new(IE_COUL) + new(IE_RECIP)
            if (decide(old, new, t, p%mtype, p%u, verdict)) then
                accepted = .true.
                if (restore) call random_seed(put=p%rng)
                if (p%mtype == TYPE_TRANSLATION) then
                    S%res(t)%com(:, m) = p%com
                else
                    S%res(t)%off(:, 1:n1, m) = p%off
                end if
                S%energy(IE_RECIP) = S%energy(IE_RECIP) + new(IE_RECIP) - old(IE_RECIP)
                S%energy(IE_NONC) = S%energy(IE_NONC) + new(IE_NONC) - old(IE_NONC)
                S%energy(IE_COUL) = S%energy(IE_COUL) + new(IE_COUL) - old(IE_COUL)
                S%energy(IE_TOTAL) = S%energy(IE_TOTAL) + new(IE_TOTAL) - old(IE_TOTAL)
                if (p%mtype == TYPE_TRANSLATION) then
                    S%counter(C_T) = S%counter(C_T) + 1
                else
                    S%counter(C_R) = S%counter(C_R) + 1
                end if
                if (verdict == BY_HOST) call window_commit(p, n_cand)
            end if
        case (TYPE_CREATION)
            S%counter(C_TRIAL_C) = S%counter(C_TRIAL_C) + 1
            old(IE_RECIP) = S%energy(IE_RECIP)
            new(1:5) = w
            old(IE_TOTAL) = old(IE_NONC) + old(IE_COUL) + old(IE_RECIP) + old(IE_SELF) + old(IE_INTRA)
            new(IE_TOTAL) = new(IE_NONC) + new(IE_COUL) + new(IE_RECIP) + new(IE_SELF) + new(IE_INTRA)
            S%res(t)%count = S%res(t)%count + 1
            S%box%num_atoms = S%box%num_atoms + n1
            if (decide(old, new, t, TYPE_CREATION, p%u, verdict)) then
                accepted = .true.
                if (restore) call random_seed(put=p%rng)
                S%res(t)%com(:, m) = p%com
                S%res(t)%off(:, 1:n1, m) = p%off
                S%energy(IE_RECIP) = new(IE_RECIP)
                S%energy(IE_NONC) = S%energy(IE_NONC) + new(IE_NONC) - old(IE_NONC)
                S%energy(IE_COUL) = S%energy(IE_COUL) + new(IE_COUL) - old(IE_COUL)
                S%energy(IE_SELF) = S%energy(IE_SELF) + new(IE_SELF) - old(IE_SELF)
                S%energy(IE_INTRA) = S%energy(IE_INTRA) + new(IE_INTRA) - old(IE_INTRA)
                S%energy(IE_TOTAL) = S%energy(IE_TOTAL) + new(IE_TOTAL) - old(IE_TOTAL)
                S%counter(C_C) = S%counter(C_C) + 1
                if (verdict == BY_HOST) call window_commit(p, n_cand)
                if (S%has_reservoir) then
                    last = S%rsv(t)%count
                    S%rsv(t)%com(:, p%pick) = S%rsv(t)%com(:, last)
                    S%rsv(t)%off(:, 1:n1, p%pick) = S%rsv(t)%off(:, 1:n1, last)
                    S%rsv(t)%count = S%rsv(t)%count - 1
                    S%rbox%num_atoms = S%rbox%num_atoms - n1
                end if
            else
                S%box%num_atoms = S%box%num_atoms - n1
                S%res(t)%count = S%res(t)%count - 1
            end if
        case (TYPE_DELETION)
            S%counter(C_TRIAL_D) = S%counter(C_TRIAL_D) + 1
            old(1:5) = o
            old(IE_RECIP) = S%energy(IE_RECIP)
            new(IE_RECIP) = w(IE_RECIP)
            if (as_written) new(IE_RECIP) = w2(IE_RECIP)                  ! creation-kind energy of the swapped-in molecule (F3)
            old(IE_TOTAL) = old(IE_NONC) + old(IE_COUL) + old(IE_RECIP) + old(IE_SELF) + old(IE_INTRA)
            new(IE_TOTAL) = new(IE_NONC) + new(IE_COUL) + new(IE_RECIP) + new(IE_SELF) + new(IE_INTRA)
            last = S%res(t)%count
            S%res(t)%count = S%res(t)%count - 1
            S%box%num_atoms = S%box%num_atoms - n1
            if (decide(old, new, t, TYPE_DELETION, p%u, verdict)) then
                accepted = .true.
                if (restore) call random_seed(put=p%rng)
                off_last = S%res(t)%off(:, :, last)
                S%res(t)%com(:, m) = S%res(t)%com(:, last)                ! RemoveMolecule: slot m <- the last slot
                S%res(t)%off(:, :, m) = S%res(t)%off(:, :, last)
                S%energy(IE_RECIP) = new(IE_RECIP)
                S%energy(IE_NONC) = S%energy(IE_NONC) + new(IE_NONC) - old(IE_NONC)
                S%energy(IE_COUL) = S%energy(IE_COUL) + new(IE_COUL) - old(IE_COUL)
                S%energy(IE_SELF) = S%energy(IE_SELF) + new(IE_SELF) - old(IE_SELF)
                S%energy(IE_INTRA) = S%energy(IE_INTRA) + new(IE_INTRA) - old(IE_INTRA)
                S%energy(IE_TOTAL) = S%energy(IE_TOTAL) + new(IE_TOTAL) - old(IE_TOTAL)
                S%counter(C_D) = S%counter(C_D) + 1
                if (verdict /= BY_HOST) then
                    continue                                                   ! the engine has applied the removal
                else if (as_written) then
                    allocate(sites_last(3, n1))
                    call MoleculeSites(S%res(t)%com(:, m), S%res(t)%off(:, 1:n1, m), n1, sites_last)
                    stat = mgpu_replica_replace_molecule(S%engine, 0_c_int, int(t - 1, c_int), int(m - 1, c_int), int(last - 1, c_int))
                    call note(stat)
                    stat = mgpu_replica_set_num_molecules(S%engine, 0_c_int, int(t - 1, c_int), int(last - 1, c_int))
                    call note(stat)
                    stat = mgpu_structure_factor_add(S%engine, 0_c_int, int(t - 1, c_int), sites_last)
                    call note(stat)
                    deallocate(sites_last)
                else
                    call window_commit(p, n_cand)
                end if
                if (S%has_reservoir) then
                    call random_number(trial)
                    trial = trial - half
                    associate (r => S%rsv(t))
                        if (r%count + 1 <= r%cap) then
                            r%com(:, r%count + 1) = trial(1) * S%rbox%matrix(:, 1) + trial(2) * S%rbox%matrix(:, 2) + &
                                                    trial(3) * S%rbox%matrix(:, 3)
                            r%off(:, 1:n1, r%count + 1) = off_last(:, 1:n1)
                            r%count = r%count + 1
                            S%rbox%num_atoms = S%rbox%num_atoms + n1
                        else
                            call note(4)
                        end if
                    end associate
                end if
            else
                S%res(t)%count = S%res(t)%count + 1
                S%box%num_atoms = S%box%num_atoms + n1
            end if
        end select
    end subroutine resolve_step

    ! the step's outcome: the engine's where it has decided, else the rule applied here
    function decide(old, new, t, move_type, u, verdict) result(yes)
        real(real64), intent(in) :: old(6), new(6), u
        integer, intent(in) :: t, move_type, verdict
        logical :: yes
        if (verdict == BY_HOST) then
            yes = u <= acceptance_probability(old, new, t, move_type)
        else
            yes = verdict == DEVICE_ACCEPTED
        end if
    end function decide

    ! apply candidate p%cand of the window's batch from the rows still resident on the lane -- or, after a one-launch
    ! window (whose rows are not the lane's), from the proposal itself
    subroutine window_commit(p, n_cand)
        type(proposal), intent(in) :: p
        integer, intent(in) :: n_cand
        integer(c_int) :: rc, rep(1), tt(1), mm(1), kk(1), acc(1)
        real(real64), allocatable, target :: sites(:, :)
        integer :: n1
        if (chain_cap > 0) then
            n1 = S%res(p%t)%n1
            allocate(sites(3, n1))
            call MoleculeSites(p%com, p%off, n1, sites)
            rep = 0; tt = p%t - 1; mm = p%m - 1; kk = int(p%kind, c_int); acc = 1
            if (p%kind == MGPU_CREATION) mm = -1
            rc = mgpu_commit_submit(S%engine, 0_c_int, 1_c_int, rep, tt, mm, kk, c_loc(sites), int(n1, c_int), acc)
            call note(int(rc))
            return
        end if
        w_acc(1:n_cand) = 0
        w_acc(p%cand) = 1
        rc = mgpu_commit_submit(S%engine, 0_c_int, int(n_cand, c_int), w_rep, w_t, w_m, w_kind, c_null_ptr, &
                                int(w_stride, c_int), w_acc)
        call note(int(rc))
    end subroutine window_commit

    ! Propose up to kwin steps, evaluate them in one engine call, walk them; returns the number of steps consumed.
    ! With the engine's one-launch path (chain_cap > 0) the engine also applies the acceptance rule to the steps in order
    ! and commits the first accepted one; the walk then only books what the engine decided.
    function run_window(kwin) result(done)
        integer, intent(in) :: kwin
        integer :: done
        type(proposal), allocatable :: P(:)
        real(real64), allocatable :: sites(:, :, :), o(:, :), w(:, :)
        real(real64) :: none5(5)
        integer :: j, n_prop, n_cand, t, n1, c, last, mx, verdict
        integer(c_int) :: rc, first, und
        logical :: accepted, one_launch
        one_launch = chain_cap > 0
        mx = 1
        do t = 1, S%n_res
            mx = max(mx, S%res(t)%n1)
        end do
        allocate(P(kwin))
        if (.not. allocated(w_rep)) then
            allocate(w_rep(128), w_t(128), w_m(128), w_kind(128), w_acc(128), w_link(128), w_u(128), w_pref(128))
        end if
        allocate(sites(3, mx, 2 * kwin), o(5, 2 * kwin), w(5, 2 * kwin))
        sites = zero
        w_stride = mx
        n_prop = 0
        n_cand = 0
        do j = 1, kwin
            ! a step may need two rows (as-written deletion): never draw a step the engine's window cannot hold
            if (one_launch .and. n_cand + 2 > chain_cap) exit
            call propose_step(P(j))
            n_prop = j
            if (P(j)%kind == STEP_ABORT) exit
            if (P(j)%kind == STEP_NOOP) cycle
            t = P(j)%t
            n1 = S%res(t)%n1
            n_cand = n_cand + 1
            P(j)%cand = n_cand
            w_rep(n_cand) = 0
            w_t(n_cand) = t - 1
            w_m(n_cand) = P(j)%m - 1
            if (P(j)%kind == MGPU_CREATION) w_m(n_cand) = -1
            w_kind(n_cand) = P(j)%kind
            w_link(n_cand) = -1
            w_u(n_cand) = P(j)%u
            select case (P(j)%mtype)
            case (TYPE_CREATION)
                w_pref(n_cand) = acceptance_prefactor(t, TYPE_CREATION, S%res(t)%count + 1)
            case (TYPE_DELETION)
                w_pref(n_cand) = acceptance_prefactor(t, TYPE_DELETION, S%res(t)%count - 1)
            case default
                w_pref(n_cand) = one
            end select
            call MoleculeSites(P(j)%com, P(j)%off, n1, sites(:, 1:n1, n_cand))
            if (P(j)%kind == MGPU_DELETION .and. as_written) then
                ! F3: the reciprocal update runs with is_creation on the molecule RemoveMolecule moves into slot m
                last = S%res(t)%count
                n_cand = n_cand + 1
                P(j)%cand2 = n_cand
                w_link(n_cand - 1) = n_cand - 1                             ! 0-based row of the companion
                w_rep(n_cand) = 0
                w_t(n_cand) = t - 1
                w_m(n_cand) = -1
                w_kind(n_cand) = MGPU_CREATION
                w_link(n_cand) = -2                                         ! energy only
                w_u(n_cand) = zero
                w_pref(n_cand) = zero
                call MoleculeSites(S%res(t)%com(:, last), S%res(t)%off(:, 1:n1, last), n1, sites(:, 1:n1, n_cand))
            end if
        end do
        first = -1
        und = -1
        if (n_cand > 0) then
            if (one_launch) then
                rc = mgpu_chain_window(S%engine, 0_c_int, int(n_cand, c_int), w_t, w_m, w_kind, w_link, sites, int(mx, c_int), &
                                       w_u, w_pref, S%temperature, S%energy(IE_RECIP), o, w, first, und)
            else
                rc = mgpu_gcmc_trial_submit(S%engine, 0_c_int, int(n_cand, c_int), w_rep, w_t, w_m, w_kind, sites, int(mx, c_int))
                if (rc == MGPU_OK) rc = mgpu_gcmc_trial_wait(S%engine, 0_c_int, o, w)
            end if
            call note(int(rc))
        end if
        none5 = zero
        done = n_prop
        do j = 1, n_prop
            if (status /= 0) then
                done = j - 1
                exit
            end if
            if (P(j)%kind == STEP_ABORT) then
                call note(4)
                done = j
                exit
            end if
            if (P(j)%kind == STEP_NOOP) cycle
            c = P(j)%cand
            verdict = BY_HOST
            if (one_launch) then
                verdict = DEVICE_REJECTED
                if (c - 1 == first) verdict = DEVICE_ACCEPTED
                if (c - 1 == und) verdict = BY_HOST                         ! too close to call on the device: decided here
            end if
            if (P(j)%cand2 > 0) then
                call resolve_step(P(j), o(:, c), w(:, c), w(:, P(j)%cand2), n_cand, j < n_prop, accepted, verdict)
            else
                call resolve_step(P(j), o(:, c), w(:, c), none5, n_cand, j < n_prop, accepted, verdict)
            end if
            if (accepted) then
                done = j
                exit
            end if
            if (one_launch .and. c - 1 == und) then
                ! rejected here: the engine decided nothing behind this step, so the window ends with it
                if (j < n_prop) call random_seed(put=P(j)%rng)
                done = j
                exit
            end if
        end do
        done = max(done, 1)
    end function run_window

    ! AdjustMoveStepSizes, as written -- including the second branches that compare against +TOL and the
    ! rotation step that is multiplied by 1.95 and clamped from above by MIN_ROTATION_ANGLE
    subroutine adjust_move_step_sizes()
        real(real64) :: acc
        if (.not. S%recalibrate) return
        if (S%counter(C_TRIAL_T) > MIN_TRIALS_FOR_RECALIBRATION) then
            acc = real(S%counter(C_T)) / real(S%counter(C_TRIAL_T))
            if (acc - TARGET_ACCEPTANCE > TOL_ACCEPTANCE) then
                S%translation_step = min(S%translation_step * 1.05d0, MAX_TRANSLATION_STEP)
            else if (acc - TARGET_ACCEPTANCE < TOL_ACCEPTANCE) then
                S%translation_step = max(S%translation_step * 0.95d0, MIN_TRANSLATION_STEP)
            end if
        end if
        if (S%counter(C_TRIAL_R) > MIN_TRIALS_FOR_RECALIBRATION) then
            acc = real(S%counter(C_R)) / real(S%counter(C_TRIAL_R))
            if (acc - TARGET_ACCEPTANCE > TOL_ACCEPTANCE) then
                S%rotation_step = min(S%rotation_step * 1.05d0, MAX_ROTATION_ANGLE)
            else if (acc - TARGET_ACCEPTANCE < TOL_ACCEPTANCE) then
                S%rotation_step = min(S%rotation_step * 1.95d0, MIN_ROTATION_ANGLE)
            end if
        end if
    end subroutine adjust_move_step_sizes

    function pick_residue_type() result(t)
        integer :: t, n_active, k, i
        n_active = 0
        do i = 1, S%n_res
            if (S%res(i)%active == 1) n_active = n_active + 1
        end do
        t = 0
        if (n_active == 0) return
        k = int(rand_uniform() * n_active) + 1
        do i = 1, S%n_res
            if (S%res(i)%active == 1) then
                k = k - 1
                if (k == 0) then
                    t = i
                    return
                end if
            end if
        end do
    end function pick_residue_type

    function pick_molecule_index(n) result(m)
        integer, intent(in) :: n
        integer :: m
        if (n == 0) then
            m = 0
        else
            m = int(rand_uniform() * n) + 1
            if (m > n) m = n
        end if
    end function pick_molecule_index

    !---------------------------------------------------------------------------
    ! program MANIAC from ComputeSystemEnergy on (main.f90:26-33): initial energy, MonteCarloLoop,
    ! FinalReport, with every output file.  outdir ends with '/'.  seed > 0 seeds the generator by the
    ! reference's rule; seed <= 0 leaves it as it is.  Returns the first engine status met (0 = none).
    !---------------------------------------------------------------------------
    function mchain_run(nb_block, nb_step, seed, outdir) bind(C, name="mchain_run") result(rc)
        integer(c_int), value :: nb_block, nb_step, seed
        character(kind=c_char), intent(in) :: outdir(*)
        integer(c_int) :: rc
        integer :: i, t, m, stat, step
        integer(c_int) :: cap
        integer(kind=8) :: c0, c1, crate
        real(real64) :: draw, e6(6)

        S%outdir = ''
        do i = 1, len(S%outdir)
            if (outdir(i) == c_null_char) exit
            S%outdir(i:i) = outdir(i)
        end do
        S%nb_block = nb_block
        S%nb_step = nb_step
        S%counter = 0
        open(unit=S%log_unit, file=trim(S%outdir) // 'log.maniac', status='replace')
        call write_log_header(S)

        call system_clock(c0, crate)
        call ComputeSystemEnergy(S%engine, e6, stat=stat)
        call note(stat)
        call system_clock(c1)
        init_seconds = real(c1 - c0, real64) / real(crate, real64)
        file_seconds = 0.0_real64
        S%energy = e6
        if (seed > 0) call seed_rng(int(seed))

        ! the engine's one-launch window path, where it applies to this system (and the host wants it)
        chain_cap = 0
        if (fused .and. chain_windows) then
            stat = mgpu_chain_window_capacity(S%engine, cap)
            call note(stat)
            if (stat == 0) chain_cap = int(cap)
        end if
        loop_seconds = 0.0_real64

        S%current_block = 0
        call log_start_mc(S)
        call system_clock(c0, crate)
        call update_files(S, .false.)
        call system_clock(c1)
        file_seconds = file_seconds + real(c1 - c0, real64) / real(crate, real64)
        do while (S%current_block < nb_block .and. status == 0)
            S%current_block = S%current_block + 1
            step = 1
            call system_clock(c0, crate)
            do while (step <= nb_step)
                if (fused .and. (spec_k > 1 .or. chain_cap > 0)) then
                    ! a window never crosses the end of a block (AdjustMoveStepSizes and the files come there)
                    step = step + run_window(min(spec_k, nb_step - step + 1))
                    if (status /= 0) exit
                    cycle
                end if
                t = pick_residue_type()
                m = pick_molecule_index(S%res(t)%count)
                draw = rand_uniform()
                if (draw <= S%p_translation) then
                    call translation(t, m)
                else if (draw <= S%p_rotation + S%p_translation) then
                    call rotation(t, m)
                else
                    if (rand_uniform() <= PROB_CREATE_DELETE) then
                        m = S%res(t)%count + 1
                        call create_molecule(t, m)
                    else
                        call delete_molecule(t, m)
                    end if
                end if
                if (status /= 0) exit
                step = step + 1
            end do
            call system_clock(c1)
            loop_seconds = loop_seconds + real(c1 - c0, real64) / real(crate, real64)
            call adjust_move_step_sizes()
            call system_clock(c0)
            call log_status(S)
            call update_files(S, .true.)
            call system_clock(c1)
            file_seconds = file_seconds + real(c1 - c0, real64) / real(crate, real64)
        end do
        call note(int(mgpu_synchronize(S%engine)))
        ! a Fortran do variable ends one past its limit: FinalReport prints nb_block + 1
        if (status == 0) S%current_block = nb_block + 1
        call log_final_report(S)
        close(S%log_unit)
        rc = status
    end function mchain_run

    !---------------------------------------------------------------------------
    ! Read-back for tests
    !---------------------------------------------------------------------------
    subroutine mchain_get_energy(e) bind(C, name="mchain_get_energy")
        real(c_double), intent(out) :: e(6)
        e = S%energy
    end subroutine mchain_get_energy

    subroutine mchain_get_counters(c) bind(C, name="mchain_get_counters")
        integer(c_int), intent(out) :: c(8)
        c = S%counter
    end subroutine mchain_get_counters

    subroutine mchain_get_counts(n) bind(C, name="mchain_get_counts")
        integer(c_int), intent(out) :: n(*)
        integer :: t
        do t = 1, S%n_res
            n(t) = S%res(t)%count
        end do
    end subroutine mchain_get_counts

    subroutine mchain_get_steps(steps) bind(C, name="mchain_get_steps")
        real(c_double), intent(out) :: steps(2)
        steps(1) = S%translation_step
        steps(2) = S%rotation_step
    end subroutine mchain_get_steps

    subroutine mchain_get_molecule(t, m, com, off) bind(C, name="mchain_get_molecule")
        integer(c_int), value :: t, m
        real(c_double), intent(out) :: com(3), off(3, *)
        com = S%res(t)%com(:, m)
        off(:, 1:S%res(t)%n1) = S%res(t)%off(:, :, m)
    end subroutine mchain_get_molecule

end module mc_chain
