!===============================================================================
! mc_farm -- Fortran host-side Metropolis driver for a farm of independent chains
! ("replicas") on one GPU.
!
! The sequential accept/reject logic stays on the host and in Fortran, with the
! reference's move set and rules:
!   move selection       src/monte_carlo.f90:50-75
!   Translation          src/translation.f90:36-112   (rand_symmetric(3)*step, ApplyPBC)
!   Rotation             src/rotation.f90:34-75, src/monte_carlo_utils.f90:30-92
!                        (theta = (u - 1/2)*rotation_step_angle about a random Cartesian axis)
!   acceptance           src/monte_carlo_utils.f90:184-226   min(1, exp(-dE/T)), energies in K
!   AcceptMove           src/monte_carlo_utils.f90:410-422
!   step recalibration   src/monte_carlo_utils.f90:99-130    (AdjustMoveStepSizes, as written)
! Random numbers: rng_kind 0 draws from the intrinsic random_number exactly like the reference
! (src/random_utils.f90:13-56); rng_kind 1 (default of the bench) uses an inlined xoshiro256+
! generator seeded by the same rule, because flang's random_number costs ~17 ns per number and the
! farm consumes nine numbers per trial (measured: 40 % of the host time of a step).
!
! One chain is sequential, so the farm advances R chains in lock step: each step
! generates one trial move per chain, evaluates all of them in one batched call
! (old and new state of every candidate), applies the Metropolis test per chain and
! commits the accepted ones.  The chains are split into two groups that alternate on
! the engine's two submission lanes, so the host prepares / resolves one group while
! the GPU evaluates the other.
!
! The per-chain loops (gathering a molecule from the host mirror, building the move, the
! Metropolis test) are independent across chains and run under OpenMP: with ~1000 chains the
! random gathers from the ~300 MB mirror are DRAM/TLB-latency bound (measured 0.34 us per chain
! single-threaded), and a few host threads overlap them.
!
! Only translation / rotation (NVT) are driven here; insertion / deletion go through
! the same engine calls (mgpu_*_candidates with MGPU_CREATION / MGPU_DELETION).
!===============================================================================
module mc_farm

    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: real64, int64
    use omp_lib
    use maniac_gpu

    implicit none

    private
    public :: mfarm_create, mfarm_run, mfarm_destroy, mfarm_get_energy, mfarm_get_molecule, mfarm_recalibrate
    public :: mfarm_get_timers

    real(real64), parameter :: PI = 3.14159265358979323846_real64
    real(real64), parameter :: TWOPI = 2.0_real64 * PI
    ! src/parameters.f90:14-21
    real(real64), parameter :: TARGET_ACCEPTANCE = 0.40d0, TOL_ACCEPTANCE = 0.05d0
    real(real64), parameter :: MIN_TRANSLATION_STEP = 1.0d-3, MAX_TRANSLATION_STEP = 3.0d0
    real(real64), parameter :: MIN_ROTATION_ANGLE = 1.0d-3, MAX_ROTATION_ANGLE = 0.78d0
    integer, parameter :: MIN_TRIALS_FOR_RECALIBRATION = 500
    integer, parameter :: NRAND = 9                       ! uniform numbers consumed per trial

    type :: lane_buffers
        integer :: first = 0, n = 0                        ! replicas [first, first + n)
        integer(c_int), allocatable :: rep(:), t(:), m(:), kind(:), accept(:)
        integer, allocatable :: ia(:)                      ! index into the active-type tables
        logical, allocatable :: is_trans(:)
        real(real64), allocatable :: sites(:, :, :)        ! (3, max_n1, n)
        real(real64), allocatable :: new_com(:, :), new_off(:, :, :)
        real(real64), allocatable :: old_e(:, :), new_e(:, :), u(:, :)
    end type lane_buffers

    type :: farm_state
        type(c_ptr) :: engine = c_null_ptr
        integer :: n_replicas = 0, n_active = 0, max_n1 = 0, cap_total = 0
        integer, allocatable :: res_type(:), n1(:), n_mol(:), first(:)   ! per active type
        real(real64), allocatable :: com(:, :, :)          ! (3, cap_total, R)    primary%mol_com
        real(real64), allocatable :: off(:, :, :, :)       ! (3, max_n1, cap_total, R) primary%site_offset
        real(real64), allocatable :: energy(:, :)          ! (3, R) non_coulomb, coulomb, recip_coulomb
        real(real64) :: lo(3), len(3), temperature, translation_step, rotation_step, p_translation
        integer(int64) :: trials = 0, accepted = 0
        integer(int64) :: trial_translations = 0, translations = 0, trial_rotations = 0, rotations = 0
        type(lane_buffers) :: lane(0:MGPU_LANES - 1)
        logical :: ready = .false.
        integer(int64) :: ticks(7) = 0                     ! generate, submit, wait, resolve, commit, rng, gather
        integer :: rng_kind = 1, n_threads = 1
        integer(int64) :: xs(4) = 0                        ! xoshiro256+ state
    end type farm_state

    type(farm_state), save, target :: F

contains

    ! seed_rng (src/random_utils.f90:33-56): seed + 37*(i-1)
    subroutine seed_farm_rng(seed)
        integer, intent(in) :: seed
        integer :: n, i
        integer, allocatable :: s(:)
        call random_seed(size=n)
        allocate(s(n))
        s = seed + 37 * [(i - 1, i = 1, n)]
        call random_seed(put=s)
        ! xoshiro256+ state from the same rule (seed + 37*(i-1)), scrambled by xorshift steps
        do i = 1, 4
            F%xs(i) = int(seed + 37 * (i - 1), int64) + 88172645463325252_int64 * int(i, int64)
            F%xs(i) = ieor(F%xs(i), ishft(F%xs(i), 13))
            F%xs(i) = ieor(F%xs(i), ishft(F%xs(i), -7))
            F%xs(i) = ieor(F%xs(i), ishft(F%xs(i), 17))
        end do
        if (all(F%xs == 0_int64)) F%xs(1) = 1_int64
    end subroutine seed_farm_rng

    ! fill u with uniform numbers in [0, 1)
    subroutine farm_random(u)
        real(real64), intent(out) :: u(:, :)
        integer :: i, j
        integer(int64) :: t, s1, s2, s3, s4
        if (F%rng_kind == 0) then
            call random_number(u)
            return
        end if
        s1 = F%xs(1); s2 = F%xs(2); s3 = F%xs(3); s4 = F%xs(4)
        do j = 1, size(u, 2)
            do i = 1, size(u, 1)
                ! xoshiro256+ (Blackman & Vigna): top 53 bits of s1 + s4
                u(i, j) = real(ishft(s1 + s4, -11), real64) * (1.0_real64 / 9007199254740992.0_real64)
                t = ishft(s2, 17)
                s3 = ieor(s3, s1); s4 = ieor(s4, s2); s2 = ieor(s2, s3); s1 = ieor(s1, s4)
                s3 = ieor(s3, t)
                s4 = ior(ishft(s4, 45), ishft(s4, -19))
            end do
        end do
        F%xs = [s1, s2, s3, s4]
    end subroutine farm_random

    !---------------------------------------------------------------------------
    ! Create the farm.  Every replica of `engine` must already hold the same configuration
    ! (mgpu_replica_copy) with A(k) initialised.  Active residue types are listed in
    ! res_type(1:n_active) (0-based engine ids); com / off hold their molecules back to back:
    !   com(3, cap_total), off(3, max_n1, cap_total), type ia occupying slots
    !   first(ia)+1 .. first(ia)+n_mol(ia).
    ! energy0 = non_coulomb, coulomb, recip_coulomb of that configuration.
    !---------------------------------------------------------------------------
    function mfarm_create(engine, n_replicas, n_active, res_type, n1, n_mol, max_n1, com, off, energy0, &
                          bounds_lo, box_len, temperature, translation_step, rotation_step, p_translation, seed, &
                          rng_kind, n_threads) bind(C, name="mfarm_create") result(rc)
        type(c_ptr), value :: engine
        integer(c_int), value :: n_replicas, n_active, max_n1, seed, rng_kind, n_threads
        integer(c_int), intent(in) :: res_type(n_active), n1(n_active), n_mol(n_active)
        real(c_double), intent(in) :: com(3, *), off(3, max_n1, *), energy0(3), bounds_lo(3), box_len(3)
        real(c_double), value :: temperature, translation_step, rotation_step, p_translation
        integer(c_int) :: rc
        integer :: ia, r, g, per, cap

        call mfarm_destroy()
        F%engine = engine
        F%n_replicas = n_replicas
        F%n_active = n_active
        F%max_n1 = max_n1
        allocate(F%res_type(n_active), F%n1(n_active), F%n_mol(n_active), F%first(n_active))
        F%res_type = res_type
        F%n1 = n1
        F%n_mol = n_mol
        cap = 0
        do ia = 1, n_active
            F%first(ia) = cap
            cap = cap + n_mol(ia)
        end do
        F%cap_total = cap
        allocate(F%com(3, cap, n_replicas), F%off(3, max_n1, cap, n_replicas), F%energy(3, n_replicas))
        do r = 1, n_replicas
            F%com(:, :, r) = com(:, 1:cap)
            F%off(:, :, :, r) = off(:, :, 1:cap)
            F%energy(:, r) = energy0
        end do
        F%lo = bounds_lo
        F%len = box_len
        F%temperature = temperature
        F%translation_step = translation_step
        F%rotation_step = rotation_step
        F%p_translation = p_translation
        F%trials = 0; F%accepted = 0
        F%trial_translations = 0; F%translations = 0; F%trial_rotations = 0; F%rotations = 0
        F%rng_kind = rng_kind
        F%n_threads = max(1, int(n_threads))
        call seed_farm_rng(int(seed))
        ! two groups of replicas, one per engine lane (a single group if there is one replica)
        per = (n_replicas + MGPU_LANES - 1) / MGPU_LANES
        do g = 0, MGPU_LANES - 1
            F%lane(g)%first = min(g * per, n_replicas)
            F%lane(g)%n = max(0, min(per, n_replicas - g * per))
            call alloc_lane(F%lane(g), max(1, F%lane(g)%n), max_n1)
        end do
        F%ready = .true.
        rc = MGPU_OK
    end function mfarm_create

    subroutine alloc_lane(L, n, max_n1)
        type(lane_buffers), intent(inout) :: L
        integer, intent(in) :: n, max_n1
        allocate(L%rep(n), L%t(n), L%m(n), L%kind(n), L%accept(n), L%ia(n), L%is_trans(n))
        allocate(L%sites(3, max_n1, n), L%new_com(3, n), L%new_off(3, max_n1, n))
        allocate(L%old_e(3, n), L%new_e(3, n), L%u(NRAND, n))
        L%sites = 0.0_real64
        L%kind = MGPU_MOVE
    end subroutine alloc_lane

    subroutine mfarm_destroy() bind(C, name="mfarm_destroy")
        integer :: g
        if (allocated(F%res_type)) deallocate(F%res_type, F%n1, F%n_mol, F%first)
        if (allocated(F%com)) deallocate(F%com, F%off, F%energy)
        do g = 0, MGPU_LANES - 1
            if (allocated(F%lane(g)%rep)) then
                deallocate(F%lane(g)%rep, F%lane(g)%t, F%lane(g)%m, F%lane(g)%kind, F%lane(g)%accept, &
                           F%lane(g)%ia, F%lane(g)%is_trans, F%lane(g)%sites, F%lane(g)%new_com, &
                           F%lane(g)%new_off, F%lane(g)%old_e, F%lane(g)%new_e, F%lane(g)%u)
            end if
        end do
        F%ready = .false.
    end subroutine mfarm_destroy

    ! RotationMatrix (src/helper_utils.f90:39-77)
    pure function RotationMatrix(axis, theta) result(r)
        integer, intent(in) :: axis
        real(real64), intent(in) :: theta
        real(real64) :: r(3, 3), c, s
        c = cos(theta)
        s = sin(theta)
        r = 0.0_real64
        r(1, 1) = 1.0_real64; r(2, 2) = 1.0_real64; r(3, 3) = 1.0_real64
        select case (axis)
        case (1)
            r(2, 2) = c; r(2, 3) = -s; r(3, 2) = s; r(3, 3) = c
        case (2)
            r(1, 1) = c; r(1, 3) = s; r(3, 1) = -s; r(3, 3) = c
        case (3)
            r(1, 1) = c; r(1, 2) = -s; r(2, 1) = s; r(2, 2) = c
        end select
    end function RotationMatrix

    !---------------------------------------------------------------------------
    ! One trial move per replica of a lane (monte_carlo.f90:50-58 + RandomTranslation /
    ! ApplyRandomRotation), then queue its evaluation.
    !---------------------------------------------------------------------------
    function generate_and_submit(g) result(rc)
        integer, intent(in) :: g
        integer(c_int) :: rc
        integer :: i, r, ia, slot, n1, axis, d, a
        integer(int64) :: c0, c1, c2, c3, c4
        integer :: p, q
        real(real64) :: theta, c, sn, x, y
        type(lane_buffers), pointer :: L
        L => F%lane(g)
        rc = MGPU_OK
        if (L%n == 0) return
        call system_clock(c0)
        call farm_random(L%u(:, 1:L%n))
        call system_clock(c3)
        ! pass 1: pick (type, molecule) and gather its com / offsets from the host mirror.  Kept free
        ! of arithmetic so the out-of-order core overlaps the cache misses of independent chains.
        !$omp parallel do num_threads(F%n_threads) schedule(static) private(r, ia, slot)
        do i = 1, L%n
            r = L%first + i                                           ! 1-based replica
            ia = min(int(L%u(1, i) * F%n_active) + 1, F%n_active)      ! PickRandomResidueType
            slot = min(int(L%u(2, i) * F%n_mol(ia)) + 1, F%n_mol(ia))  ! PickRandomMoleculeIndex
            L%ia(i) = ia
            L%rep(i) = r - 1
            L%t(i) = F%res_type(ia)
            L%m(i) = slot - 1
            L%new_com(:, i) = F%com(:, F%first(ia) + slot, r)
            L%new_off(:, :, i) = F%off(:, :, F%first(ia) + slot, r)
        end do
        !$omp end parallel do
        call system_clock(c4)
        F%ticks(6) = F%ticks(6) + (c3 - c0)
        F%ticks(7) = F%ticks(7) + (c4 - c3)
        ! pass 2: the moves themselves
        !$omp parallel do num_threads(F%n_threads) schedule(static) private(n1, d, x, y, theta, axis, c, sn, p, q, a)
        do i = 1, L%n
            n1 = F%n1(L%ia(i))
            L%is_trans(i) = (L%u(3, i) <= F%p_translation) .or. (n1 == 1)
            if (L%is_trans(i)) then
                ! translation.f90:104-110: rand_symmetric(3)*translation_step, then ApplyPBC
                ! (geometry_utils.f90:190: lo + modulo(pos - lo, L); written out: pos - lo is
                !  within one box length of [0, L) for any legal step)
                do d = 1, 3
                    x = (L%new_com(d, i) + (L%u(3 + d, i) - 0.5_real64) * F%translation_step) - F%lo(d)
                    if (x < 0.0_real64 .or. x >= F%len(d)) x = modulo(x, F%len(d))
                    L%new_com(d, i) = F%lo(d) + x
                end do
            else
                ! monte_carlo_utils.f90:54-64 with RotationMatrix (helper_utils.f90:39-77) written out:
                ! rotation by theta about Cartesian axis `axis` mixes the two other components
                theta = (L%u(7, i) - 0.5_real64) * F%rotation_step
                axis = int(L%u(8, i) * 3.0_real64) + 1
                c = cos(theta)
                sn = sin(theta)
                p = mod(axis, 3) + 1          ! X -> (Y, Z), Y -> (Z, X), Z -> (X, Y)
                q = mod(axis + 1, 3) + 1
                do a = 1, n1
                    x = L%new_off(p, a, i)
                    y = L%new_off(q, a, i)
                    L%new_off(p, a, i) = c * x - sn * y
                    L%new_off(q, a, i) = sn * x + c * y
                end do
            end if
            do a = 1, n1
                L%sites(:, a, i) = L%new_com(:, i) + L%new_off(:, a, i)
            end do
        end do
        !$omp end parallel do
        call system_clock(c1)
        rc = mgpu_trial_submit(F%engine, int(g, c_int), int(L%n, c_int), L%rep, L%t, L%m, L%sites, &
                               int(F%max_n1, c_int))
        call system_clock(c2)
        F%ticks(1) = F%ticks(1) + (c1 - c0)
        F%ticks(2) = F%ticks(2) + (c2 - c1)
    end function generate_and_submit

    !---------------------------------------------------------------------------
    ! Collect a lane's energies, apply the Metropolis test per replica
    ! (mc_acceptance_probability, monte_carlo_utils.f90:204-218), update the host mirrors
    ! and running energies (AcceptMove) and queue the commit of the accepted moves.
    !---------------------------------------------------------------------------
    function resolve_and_commit(g) result(rc)
        integer, intent(in) :: g
        integer(c_int) :: rc
        integer :: i, r, ia, slot, n1
        integer(int64) :: c0, c1, c2, c3, n_tt, n_t, n_rr, n_r
        real(real64) :: delta_e, probability
        type(lane_buffers), pointer :: L
        L => F%lane(g)
        rc = MGPU_OK
        if (L%n == 0) return
        call system_clock(c0)
        rc = mgpu_trial_wait(F%engine, int(g, c_int), L%old_e, L%new_e)
        if (rc /= MGPU_OK) return
        call system_clock(c1)
        n_tt = 0; n_t = 0; n_rr = 0; n_r = 0
        !$omp parallel do num_threads(F%n_threads) schedule(static) private(r, delta_e, probability, ia, slot, n1) &
        !$omp& reduction(+:n_tt, n_t, n_rr, n_r)
        do i = 1, L%n
            r = L%first + i
            delta_e = (L%new_e(1, i) + L%new_e(2, i) + L%new_e(3, i)) - (L%old_e(1, i) + L%old_e(2, i) + L%old_e(3, i))
            probability = min(1.0_real64, exp(-delta_e / F%temperature))
            if (L%is_trans(i)) then
                n_tt = n_tt + 1
            else
                n_rr = n_rr + 1
            end if
            if (L%u(9, i) <= probability) then
                L%accept(i) = 1
                ia = L%ia(i)
                slot = L%m(i) + 1
                n1 = F%n1(ia)
                F%com(:, F%first(ia) + slot, r) = L%new_com(:, i)
                F%off(:, 1:n1, F%first(ia) + slot, r) = L%new_off(:, 1:n1, i)
                F%energy(:, r) = F%energy(:, r) + L%new_e(:, i) - L%old_e(:, i)
                if (L%is_trans(i)) then
                    n_t = n_t + 1
                else
                    n_r = n_r + 1
                end if
            else
                L%accept(i) = 0
            end if
        end do
        !$omp end parallel do
        F%trial_translations = F%trial_translations + n_tt
        F%translations = F%translations + n_t
        F%trial_rotations = F%trial_rotations + n_rr
        F%rotations = F%rotations + n_r
        F%accepted = F%accepted + n_t + n_r
        F%trials = F%trials + L%n
        call system_clock(c2)
        rc = mgpu_commit_submit(F%engine, int(g, c_int), int(L%n, c_int), L%rep, L%t, L%m, L%kind, &
                                c_null_ptr, int(F%max_n1, c_int), L%accept)
        call system_clock(c3)
        F%ticks(3) = F%ticks(3) + (c1 - c0)
        F%ticks(4) = F%ticks(4) + (c2 - c1)
        F%ticks(5) = F%ticks(5) + (c3 - c2)
    end function resolve_and_commit

    !---------------------------------------------------------------------------
    ! Advance every chain by n_steps trial moves.  out = trials, accepted,
    ! trial_translations, translations, trial_rotations, rotations (cumulative).
    !---------------------------------------------------------------------------
    function mfarm_run(n_steps, out) bind(C, name="mfarm_run") result(rc)
        integer(c_int), value :: n_steps
        real(c_double), intent(out) :: out(6)
        integer(c_int) :: rc
        integer :: step, g
        rc = MGPU_OK
        out = 0.0_real64
        if (.not. F%ready) then
            rc = 5
            return
        end if
        if (n_steps > 0) then
            do g = 0, MGPU_LANES - 1
                rc = generate_and_submit(g)
                if (rc /= MGPU_OK) return
            end do
            do step = 1, n_steps
                do g = 0, MGPU_LANES - 1
                    rc = resolve_and_commit(g)
                    if (rc /= MGPU_OK) return
                    if (step < n_steps) then
                        rc = generate_and_submit(g)
                        if (rc /= MGPU_OK) return
                    end if
                end do
            end do
            rc = mgpu_synchronize(F%engine)
        end if
        out(1) = real(F%trials, real64); out(2) = real(F%accepted, real64)
        out(3) = real(F%trial_translations, real64); out(4) = real(F%translations, real64)
        out(5) = real(F%trial_rotations, real64); out(6) = real(F%rotations, real64)
    end function mfarm_run

    ! AdjustMoveStepSizes (src/monte_carlo_utils.f90:99-130), as written in the reference
    ! (including its min(..*1.95, MIN_ROTATION_ANGLE) branch), on the farm-wide counters.
    subroutine mfarm_recalibrate(steps) bind(C, name="mfarm_recalibrate")
        real(c_double), intent(out) :: steps(2)
        real(real64) :: acc
        if (F%trial_translations > MIN_TRIALS_FOR_RECALIBRATION) then
            acc = real(F%translations, real64) / real(F%trial_translations, real64)
            if (acc - TARGET_ACCEPTANCE > TOL_ACCEPTANCE) then
                F%translation_step = min(F%translation_step * 1.05d0, MAX_TRANSLATION_STEP)
            else if (acc - TARGET_ACCEPTANCE < TOL_ACCEPTANCE) then
                F%translation_step = max(F%translation_step * 0.95d0, MIN_TRANSLATION_STEP)
            end if
        end if
        if (F%trial_rotations > MIN_TRIALS_FOR_RECALIBRATION) then
            acc = real(F%rotations, real64) / real(F%trial_rotations, real64)
            if (acc - TARGET_ACCEPTANCE > TOL_ACCEPTANCE) then
                F%rotation_step = min(F%rotation_step * 1.05d0, MAX_ROTATION_ANGLE)
            else if (acc - TARGET_ACCEPTANCE < TOL_ACCEPTANCE) then
                F%rotation_step = min(F%rotation_step * 1.95d0, MIN_ROTATION_ANGLE)
            end if
        end if
        steps(1) = F%translation_step
        steps(2) = F%rotation_step
    end subroutine mfarm_recalibrate

    ! host wall time spent in: trial generation, trial submit, waiting for the GPU, Metropolis
    ! resolution, commit submit (seconds, cumulative)
    subroutine mfarm_get_timers(t) bind(C, name="mfarm_get_timers")
        real(c_double), intent(out) :: t(7)
        integer(int64) :: rate
        call system_clock(count_rate=rate)
        t = real(F%ticks, real64) / real(rate, real64)
    end subroutine mfarm_get_timers

    ! running energies (non_coulomb, coulomb, recip_coulomb) of one replica (0-based)
    subroutine mfarm_get_energy(replica, e) bind(C, name="mfarm_get_energy")
        integer(c_int), value :: replica
        real(c_double), intent(out) :: e(3)
        e = F%energy(:, replica + 1)
    end subroutine mfarm_get_energy

    ! host mirror of one molecule: active-type index ia (0-based), slot (0-based), replica (0-based)
    subroutine mfarm_get_molecule(replica, ia, slot, com, off) bind(C, name="mfarm_get_molecule")
        integer(c_int), value :: replica, ia, slot
        real(c_double), intent(out) :: com(3), off(3, F%max_n1)
        com = F%com(:, F%first(ia + 1) + slot + 1, replica + 1)
        off = F%off(:, :, F%first(ia + 1) + slot + 1, replica + 1)
    end subroutine mfarm_get_molecule

end module mc_farm
