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
!   CreateMolecule       src/create_molecule.f90:41-207 (uniform position, geometry of molecule 1,
!                        full random rotation; slot num_residues + 1)
!   DeleteMolecule       src/delete_molecule.f90:41-116 (swap-with-last)
!   acceptance           src/monte_carlo_utils.f90:184-226   energies in K, fugacity in molecules/A^3
!   AcceptMove & co.     src/monte_carlo_utils.f90:410-422, create_molecule.f90:96-131,
!                        delete_molecule.f90:126-168
!   step recalibration   src/monte_carlo_utils.f90:99-130    (AdjustMoveStepSizes, as written)
! Deliberate differences from the reference (SURVEY F2 / F3): A(k) is initialised to S(k) before
! the first move, and a deletion's new reciprocal energy is sum ff W |A - S_mol|^2.
!
! Random numbers: rng_kind 0 draws from the intrinsic random_number like the reference
! (src/random_utils.f90:13-56), serially; rng_kind 1 (default) gives every chain its own xoshiro256+
! generator, seeded from the reference's rule and the chain index: flang's random_number costs ~17 ns
! per number (ten per trial), and chain-local generators make a chain's trajectory independent of how
! many chains run beside it and let the draw run inside the parallel loop.
!
! One chain is sequential, so the farm advances R chains in lock step: each step generates one
! trial move per chain, evaluates all of them in one batched call (old and new state of every
! candidate), applies the Metropolis test per chain and commits the accepted ones.  The chains are
! split into n_lanes groups (default two) that alternate on the engine's submission lanes, so the
! host prepares / resolves one group while the GPU evaluates the others.
!
! The per-chain loops (gathering a molecule from the host mirror, building the move, the
! Metropolis test) are independent across chains and run under OpenMP: with ~1000 chains the
! random gathers from the ~300 MB mirror are DRAM/TLB-latency bound (measured 0.34 us per chain
! single-threaded), and a few host threads overlap them.
!===============================================================================
module mc_farm

    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: real64, int64
    use omp_lib
    use maniac_gpu

    implicit none

    private
    public :: mfarm_create, mfarm_run, mfarm_destroy, mfarm_get_energy, mfarm_get_molecule, mfarm_recalibrate
    public :: mfarm_get_timers, mfarm_set_gcmc, mfarm_get_counts, mfarm_get_counters, mfarm_set_triclinic
    public :: mfarm_rng_sample, mfarm_set_drivers, mfarm_configure, mfarm_select, mfarm_exchange_block
    public :: mfarm_window_mode, mfarm_set_window_depth

    real(real64), parameter :: PI = 3.14159265358979323846_real64
    real(real64), parameter :: TWOPI = 2.0_real64 * PI
    ! src/parameters.f90:14-21
    real(real64), parameter :: TARGET_ACCEPTANCE = 0.40d0, TOL_ACCEPTANCE = 0.05d0
    real(real64), parameter :: MIN_TRANSLATION_STEP = 1.0d-3, MAX_TRANSLATION_STEP = 3.0d0
    real(real64), parameter :: MIN_ROTATION_ANGLE = 1.0d-3, MAX_ROTATION_ANGLE = 0.78d0
    real(real64), parameter :: PROB_CREATE_DELETE = 0.5d0
    integer, parameter :: MIN_TRIALS_FOR_RECALIBRATION = 500
    integer, parameter :: NRAND = 10                      ! uniform numbers consumed per trial
    integer, parameter :: RNG_BLOCK = 256                 ! chains per mgpu_rng_fill call
    ! internal move codes; engine kinds are MGPU_MOVE / MGPU_CREATION / MGPU_DELETION
    integer, parameter :: MV_TRANSLATION = 1, MV_ROTATION = 2, MV_CREATION = 3, MV_DELETION = 4

    type :: lane_buffers
        integer :: first = 0, n = 0                        ! replicas [first, first + n)
        integer :: nc = 0                                  ! candidates of the trial in flight (<= n)
        integer(c_int), allocatable :: rep(:), t(:), m(:), kind(:), accept(:)
        integer, allocatable :: ia(:), move(:), cidx(:)    ! active-type index, move code, chain index
        integer, allocatable :: sel_ia(:), sel_mv(:), sel_slot(:)   ! per chain: the selection (slot 0 = no-op)
        ! (3, max_n1, n) candidate rows: built straight in the engine's pinned staging block of the lane
        ! (mgpu_lane_site_buffer), or in sites_own where that is not available
        real(real64), pointer :: sites(:, :, :) => null()
        real(real64), allocatable :: sites_own(:, :, :)
        real(real64), allocatable :: new_com(:, :), new_off(:, :, :)
        integer(c_int), allocatable :: mvc(:)              ! device-built trials: move code per candidate
        real(real64), allocatable :: u5(:, :)              ! ... and its five uniform numbers (5, n)
        real(real64), allocatable :: acc_u(:), acc_pref(:) ! device-decided trials: the test's number and prefactor
        real(real64), allocatable :: old_e(:), new_e(:)    ! (ne * nc) rows packed by the engine
        real(real64), allocatable :: u(:, :)
        ! ---- window mode (mfarm_configure(3)): one launch per lane step, up to F%depth windows of the lane in flight.
        ! A window carries one record per chain of the lane; w_live = 1 where the record is a step of the chain (0: filler).
        ! Per chain: issued / resolved count its steps of the current mfarm_run; pend_* is a FIFO of draws to send (again)
        ! before any fresh one -- an UNDECIDED step with the host's decision (pend_forced, its front entry), then the steps
        ! whose windows were already in flight behind it and did nothing.
        integer :: w_head = 0, w_count = 0
        integer, allocatable :: w_live(:, :), w_ia(:, :), w_mv(:, :), w_slot(:, :), w_forced(:, :)    ! (n, 0:depth-1)
        real(real64), allocatable :: w_u(:, :, :)                                                     ! (NRAND, n, 0:depth-1)
        integer, allocatable :: issued(:), resolved(:), pend_n(:), pend_forced(:)
        real(real64), allocatable :: pend_u(:, :, :)                                                  ! (NRAND, depth + 1, n)
        integer(c_int), allocatable :: forced(:), verdict(:)
        real(c_double), allocatable :: slot_u(:)                  ! insertion / deletion farms: the draw the engine picks the molecule with
        integer(int64) :: outstanding = 0, in_flight = 0          ! steps of the run not yet resolved / records in flight
        integer(int64) :: undecided = 0                           ! steps the device left to the driver (folded into the farm's)
        ! per-lane accumulators (lanes may run on different host threads), folded into the farm's totals by mfarm_run
        integer(int64) :: trials = 0, accepted = 0, skipped = 0, counters(8) = 0, ticks(7) = 0
    end type lane_buffers

    type :: farm_state
        type(c_ptr) :: engine = c_null_ptr
        integer :: n_replicas = 0, n_active = 0, max_n1 = 0, cap_total = 0
        integer, allocatable :: res_type(:), n1(:), cap(:), first(:)     ! per active type
        integer, allocatable :: cnt(:, :)                  ! (n_active, R)  primary%num_residues
        ! host mirror of primary%mol_com / primary%site_offset: one contiguous record per molecule,
        ! mol(1:3) = com, mol(3a+1:3a+3) = offset of site a -- one cache / TLB miss per random gather
        real(real64), pointer :: mol(:, :, :) => null()    ! (3 + 3*max_n1, cap_total, R); 2 MiB-aligned, MADV_HUGEPAGE
        type(c_ptr) :: mol_raw = c_null_ptr                ! its allocation (posix_memalign)
        real(real64), allocatable :: energy(:, :)          ! (5, R) non_coulomb, coulomb, recip, self, intra
        real(real64), allocatable :: fugacity(:, :)        ! (n_active, R) molecules per cubic Angstrom
        real(real64) :: lo(3), len(3), volume, temperature, translation_step, rotation_step
        logical :: triclinic = .false.                     ! box%is_triclinic: wrap / insert through the cell matrix
        real(real64) :: matrix(3, 3) = 0.0_real64, reciprocal(3, 3) = 0.0_real64
        real(real64) :: p_translation, p_rotation          ! the rest is insertion / deletion
        logical :: gcmc = .false.
        integer(int64) :: trials = 0, accepted = 0
        integer(int64) :: counters(8) = 0   ! trial/accepted: translations, rotations, creations, deletions
        integer(int64) :: skipped = 0                      ! no-op selections (empty type, full type)
        type(lane_buffers) :: lane(0:MGPU_LANES - 1)
        logical :: ready = .false.
        integer(int64) :: ticks(7) = 0                     ! generate, submit, wait, resolve, commit, rng, gather
        integer :: rng_kind = 1, n_threads = 1, n_lanes = 2
        integer :: team = 1                                ! OpenMP threads of one lane's loops
        logical :: lane_threads = .false.                  ! several driver threads (each with its team) share the lanes
        integer :: n_drivers = 1                           ! driver threads asked for (mfarm_set_drivers; MFARM_LANE_THREADS overrides)
        ! .true.: the engine holds the molecules' frames (mgpu_replica_set_frames) and BUILDS the trial geometry itself
        ! (mgpu_move_trial_submit): the host keeps counts, energies and the random stream -- no mirror of the coordinates,
        ! no gathers from it, no candidate rows to stage
        logical :: device_build = .false.
        logical :: device_accept = .false.       ! ... and the engine applies the acceptance rule and commits (mfarm_configure(2))
        ! window mode (mfarm_configure(3)): ONE launch per lane step (mgpu_farm_window_submit): the engine builds, evaluates,
        ! decides with the driver's draws and commits; the driver selects the moves, checks every decision against its own
        ! rule and decides itself whatever the device left undecided.  depth: windows of a lane in flight (1 with
        ! insertion / deletion: the selection then depends on the last outcome)
        logical :: window = .false.
        integer :: depth = 2
        integer(int64) :: undecided = 0
        integer(int64), allocatable :: cxs(:, :)           ! (4, R) xoshiro256+ state of every chain
    end type farm_state

    ! Several farms may exist in one process (one per engine: different boxes, force fields or GPUs); the entry points act
    ! on the SELECTED one (mfarm_select; slot 0 until told otherwise).  The selection is process-wide: farms are driven
    ! one call at a time.
    integer, parameter :: MAX_FARMS = 8
    type(farm_state), save, target :: farms(0:MAX_FARMS - 1)
    type(farm_state), pointer, save :: F => farms(0)
    logical, save :: want_device_build = .false.           ! consumed by the next mfarm_create (mfarm_configure)
    logical, save :: want_device_accept = .false.
    logical, save :: want_window = .false.

    interface
        function c_posix_memalign(ptr, alignment, bytes) bind(C, name="posix_memalign") result(rc)
            import :: c_ptr, c_size_t, c_int
            type(c_ptr), intent(out) :: ptr
            integer(c_size_t), value :: alignment, bytes
            integer(c_int) :: rc
        end function
        function c_madvise(addr, bytes, advice) bind(C, name="madvise") result(rc)
            import :: c_ptr, c_size_t, c_int
            type(c_ptr), value :: addr
            integer(c_size_t), value :: bytes
            integer(c_int), value :: advice
            integer(c_int) :: rc
        end function
        subroutine c_free(ptr) bind(C, name="free")
            import :: c_ptr
            type(c_ptr), value :: ptr
        end subroutine
    end interface

contains

    ! The host mirror is several hundred MB of 96-byte records gathered at random: with 4 KiB pages every gather
    ! also misses the TLB and walks a page table that is itself out of cache.  Ask for transparent huge pages
    ! (the GPU boxes run THP in madvise mode); a refusal changes nothing but speed.
    subroutine alloc_mirror(n_rec, n_slots, n_rep)
        integer, intent(in) :: n_rec, n_slots, n_rep
        integer(c_size_t) :: bytes
        integer(c_int) :: rc
        integer(c_int), parameter :: MADV_HUGEPAGE = 14
        integer(c_size_t), parameter :: HUGE = 2097152_c_size_t
        bytes = int(n_rec, c_size_t) * int(n_slots, c_size_t) * int(n_rep, c_size_t) * 8_c_size_t
        bytes = ((bytes + HUGE - 1) / HUGE) * HUGE
        rc = c_posix_memalign(F%mol_raw, HUGE, bytes)
        if (rc /= 0 .or. .not. c_associated(F%mol_raw)) error stop "mc_farm: cannot allocate the host mirror"
        rc = c_madvise(F%mol_raw, bytes, MADV_HUGEPAGE)
        call c_f_pointer(F%mol_raw, F%mol, [n_rec, n_slots, n_rep])
    end subroutine alloc_mirror


    ! seed_rng (src/random_utils.f90:33-56): seed + 37*(i-1) for the intrinsic generator; every chain
    ! additionally owns a xoshiro256+ generator (Blackman & Vigna) seeded from its own splitmix64 stream
    ! (mgpu_rng_seed_streams: the authors' recommended seeding, in unsigned 64-bit arithmetic on the C side), so
    ! the chains' streams are statistically independent and a chain's random numbers do not depend on how many
    ! chains run beside it
    subroutine seed_farm_rng(seed, n_chains)
        integer, intent(in) :: seed, n_chains
        integer :: n, i
        integer, allocatable :: s(:)
        call random_seed(size=n)
        allocate(s(n))
        s = seed + 37 * [(i - 1, i = 1, n)]
        call random_seed(put=s)
        call seed_chain_streams(seed, n_chains)
    end subroutine seed_farm_rng

    ! the chains' own xoshiro256+ states only (the intrinsic generator is left alone)
    subroutine seed_chain_streams(seed, n_chains)
        integer, intent(in) :: seed, n_chains
        integer(c_int) :: rc
        if (allocated(F%cxs)) deallocate(F%cxs)
        allocate(F%cxs(4, n_chains))
        rc = mgpu_rng_seed_streams(int(seed, c_long_long), int(n_chains, c_int), F%cxs)
    end subroutine seed_chain_streams

    ! NRAND uniform numbers in [0, 1) from chain r's generator: top 53 bits of (s1 + s4) mod 2^64.  The sum is
    ! formed without relying on signed overflow: q = floor((s1 + s4) / 4) from the operands' upper 62 bits plus the
    ! carry of their two low bits (q < 2^63), of which bits 9..61 are bits 11..63 of the 64-bit sum.
    subroutine chain_random(r, u)
        integer, intent(in) :: r
        real(real64), intent(out) :: u(NRAND)
        integer :: i
        integer(int64) :: t, s1, s2, s3, s4, q
        integer(int64), parameter :: MASK62 = 4611686018427387903_int64          ! 2^62 - 1
        s1 = F%cxs(1, r); s2 = F%cxs(2, r); s3 = F%cxs(3, r); s4 = F%cxs(4, r)
        do i = 1, NRAND
            q = ishft(s1, -2) + ishft(s4, -2) + ishft(iand(s1, 3_int64) + iand(s4, 3_int64), -2)
            u(i) = real(ishft(iand(q, MASK62), -9), real64) * (1.0_real64 / 9007199254740992.0_real64)
            t = ishft(s2, 17)
            s3 = ieor(s3, s1); s4 = ieor(s4, s2); s2 = ieor(s2, s3); s1 = ieor(s1, s4)
            s3 = ieor(s3, t)
            s4 = ior(ishft(s4, 45), ishft(s4, -19))
        end do
        F%cxs(1, r) = s1; F%cxs(2, r) = s2; F%cxs(3, r) = s3; F%cxs(4, r) = s4
    end subroutine chain_random

    !---------------------------------------------------------------------------
    ! Create the farm.  Every replica of `engine` must already hold the same configuration
    ! (mgpu_replica_copy) with A(k) initialised.  Active residue types are listed in
    ! res_type(1:n_active) (0-based engine ids) with n_mol molecules each and room for cap (the
    ! engine's mol_capacity); com / off hold the initial molecules of the types back to back:
    !   com(3, sum n_mol), off(3, max_n1, sum n_mol).
    ! energy0 = non_coulomb, coulomb, recip_coulomb, ewald_self, intra_coulomb of that configuration.
    !---------------------------------------------------------------------------
    function mfarm_create(engine, n_replicas, n_active, res_type, n1, n_mol, cap, max_n1, com, off, energy0, &
                          bounds_lo, box_len, temperature, translation_step, rotation_step, p_translation, seed, &
                          rng_kind, n_threads, n_lanes) bind(C, name="mfarm_create") result(rc)
        type(c_ptr), value :: engine
        integer(c_int), value :: n_replicas, n_active, max_n1, seed, rng_kind, n_threads, n_lanes
        integer(c_int), intent(in) :: res_type(n_active), n1(n_active), n_mol(n_active), cap(n_active)
        real(c_double), intent(in) :: com(3, *), off(3, max_n1, *), energy0(5), bounds_lo(3), box_len(3)
        real(c_double), value :: temperature, translation_step, rotation_step, p_translation
        integer(c_int) :: rc
        integer :: ia, r, g, per, tot, src, k, i, lane_first, lane_n

        call mfarm_destroy()
        rc = MGPU_OK
        do ia = 1, n_active
            if (n_mol(ia) > cap(ia) .or. cap(ia) < 1) then
                rc = 3
                return
            end if
        end do
        F%engine = engine
        F%n_replicas = n_replicas
        F%n_active = n_active
        F%max_n1 = max_n1
        allocate(F%res_type(n_active), F%n1(n_active), F%cap(n_active), F%first(n_active))
        allocate(F%cnt(n_active, n_replicas), F%fugacity(n_active, n_replicas))
        F%res_type = res_type
        F%n1 = n1
        F%cap = cap
        tot = 0
        do ia = 1, n_active
            F%first(ia) = tot
            tot = tot + cap(ia)
        end do
        F%cap_total = tot
        F%device_build = want_device_build
        F%device_accept = want_device_build .and. want_device_accept
        F%window = .false.
        F%undecided = 0                      ! (a slot is reused by the next farm)
        if (want_device_build .and. want_window) then
            ! where the one-launch path does not apply (triclinic box, large molecules) the farm stays on the batched path
            rc = mgpu_farm_window_capacity(engine, k, i)
            g = n_lanes
            if (g <= 0) g = 2
            g = max(1, min(g, int(MGPU_LANES), int(n_replicas)))
            F%window = rc == MGPU_OK .and. k >= (n_replicas + g - 1) / g        ! a lane's chains fit one launch
            rc = MGPU_OK
        end if
        if (.not. F%device_build) call alloc_mirror(3 + 3 * max_n1, tot, int(n_replicas))
        allocate(F%energy(5, n_replicas))
        F%n_threads = max(1, int(n_threads))
        ! n_lanes groups of replicas, one per engine lane (<= 0: the default of two)
        F%n_lanes = n_lanes
        if (n_lanes <= 0) F%n_lanes = 2
        F%n_lanes = max(1, min(F%n_lanes, int(MGPU_LANES), int(n_replicas)))
        per = (n_replicas + F%n_lanes - 1) / F%n_lanes
        ! First touch of the mirror (several hundred MB) with the loop shape of generate_and_submit -- a static
        ! split of each lane's chains -- so that a thread gathers from pages on its own NUMA node.
        do g = 0, F%n_lanes - 1
            lane_first = min(g * per, int(n_replicas))
            lane_n = max(0, min(per, int(n_replicas) - g * per))
            !$omp parallel do num_threads(F%n_threads) schedule(static) private(r, ia, k, src)
            do i = 1, lane_n
                r = lane_first + i
                if (.not. F%device_build) F%mol(:, :, r) = 0.0_real64
                src = 0
                do ia = 1, n_active
                    if (.not. F%device_build) then
                        do k = 1, n_mol(ia)
                            F%mol(1:3, F%first(ia) + k, r) = com(:, src + k)
                            F%mol(4:, F%first(ia) + k, r) = reshape(off(:, :, src + k), [3 * max_n1])
                        end do
                    end if
                    F%cnt(ia, r) = n_mol(ia)
                    src = src + n_mol(ia)
                end do
            end do
            !$omp end parallel do
        end do
        do r = 1, n_replicas
            F%energy(:, r) = energy0
        end do
        F%fugacity = 0.0_real64
        F%lo = bounds_lo
        F%len = box_len
        F%volume = box_len(1) * box_len(2) * box_len(3)
        F%triclinic = .false.
        F%temperature = temperature
        F%translation_step = translation_step
        F%rotation_step = rotation_step
        F%p_translation = p_translation
        F%p_rotation = 1.0_real64 - p_translation
        F%gcmc = .false.
        F%trials = 0; F%accepted = 0; F%counters = 0; F%skipped = 0; F%ticks = 0
        F%rng_kind = rng_kind
        F%n_threads = max(1, int(n_threads))
        F%n_drivers = 1
        call seed_farm_rng(int(seed), int(n_replicas))
        do g = 0, F%n_lanes - 1
            F%lane(g)%first = min(g * per, n_replicas)
            F%lane(g)%n = max(0, min(per, n_replicas - g * per))
            F%lane(g)%nc = 0
            call alloc_lane(F%lane(g), max(1, F%lane(g)%n), int(max_n1), g)
        end do
        F%ready = .true.
    end function mfarm_create

    !---------------------------------------------------------------------------
    ! Triclinic box (after mfarm_create): box%matrix and box%reciprocal as the Fortran arrays, and the cell
    ! volume; translations then wrap through fractional coordinates and insertions are placed with the
    ! cell matrix, as ApplyPBC / InsertAndOrientMolecule do for box%is_triclinic.
    !---------------------------------------------------------------------------
    subroutine mfarm_set_triclinic(matrix, reciprocal, volume) bind(C, name="mfarm_set_triclinic")
        real(c_double), intent(in) :: matrix(3, 3), reciprocal(3, 3)
        real(c_double), value :: volume
        F%matrix = matrix
        F%reciprocal = reciprocal
        F%volume = volume
        F%triclinic = .true.
    end subroutine mfarm_set_triclinic

    !---------------------------------------------------------------------------
    ! Switch insertion / deletion on: move probabilities as in the .maniac input
    ! (translation_proba, rotation_proba; the remainder is insertion/deletion, split 50/50,
    ! monte_carlo.f90:53-75) and one fugacity per (active type, replica), already converted to
    ! molecules per cubic Angstrom (ConvertFugacity, prepare_utils.f90:48-73).
    !---------------------------------------------------------------------------
    function mfarm_set_gcmc(p_translation, p_rotation, fugacity) bind(C, name="mfarm_set_gcmc") result(rc)
        real(c_double), value :: p_translation, p_rotation
        real(c_double), intent(in) :: fugacity(F%n_active, F%n_replicas)
        integer(c_int) :: rc
        rc = MGPU_OK
        if (.not. F%ready) then
            rc = 5
            return
        end if
        if (p_translation < 0.0_real64 .or. p_rotation < 0.0_real64 .or. &
            p_translation + p_rotation > 1.0_real64 + 1.0d-12 .or. any(fugacity <= 0.0_real64)) then
            rc = 1
            return
        end if
        F%p_translation = p_translation
        F%p_rotation = p_rotation
        F%fugacity = fugacity
        F%gcmc = .true.
    end function mfarm_set_gcmc

    ! The one exchange step of a farm of replicas spread over the GPUs of a node (SURVEY 8(e)): at a block's end every rank
    ! contributes {moves accepted, trials} so far and, per active residue type, the histogram of its chains' molecule counts
    ! -- what the reference records per chain in number_<res>.dat (src/write_utils.f90:144-150) -- and receives the
    ! rank-ordered tables: sums_by_rank(2, world), hist_by_rank(n_bins, n_active, world).  `comm` comes from
    ! mgpu_comm_create (one rank: the identity, no RCCL).  Counts beyond the last bin land in it.
    function mfarm_exchange_block(comm, n_bins, world, sums_by_rank, hist_by_rank) bind(C, name="mfarm_exchange_block") result(rc)
        type(c_ptr), value :: comm
        integer(c_int), value :: n_bins, world
        real(c_double), intent(out) :: sums_by_rank(2, world)
        integer(c_long_long), intent(out) :: hist_by_rank(n_bins, F%n_active, world)
        integer(c_int) :: rc
        integer(c_long_long), allocatable :: hist(:, :)
        real(c_double) :: sums(2)
        integer :: r, ia, b
        rc = MGPU_OK
        if (.not. F%ready .or. n_bins < 1 .or. world < 1) then
            rc = 1
            return
        end if
        allocate(hist(n_bins, F%n_active))
        hist = 0
        do r = 1, F%n_replicas
            do ia = 1, F%n_active
                b = min(max(F%cnt(ia, r), 0), n_bins - 1) + 1
                hist(b, ia) = hist(b, ia) + 1
            end do
        end do
        sums = [real(F%accepted, c_double), real(F%trials, c_double)]
        rc = mgpu_allgather_block_stats(comm, 2_c_int, sums, int(n_bins * F%n_active, c_int), hist, sums_by_rank, hist_by_rank)
    end function mfarm_exchange_block

    subroutine alloc_lane(L, n, max_n1, g)
        type(lane_buffers), intent(inout), target :: L
        integer, intent(in) :: n, max_n1, g
        type(c_ptr) :: staged
        integer(c_int) :: rc
        allocate(L%rep(n), L%t(n), L%m(n), L%kind(n), L%accept(n), L%ia(n), L%move(n), L%cidx(n))
        allocate(L%sel_ia(n), L%sel_mv(n), L%sel_slot(n))
        allocate(L%new_com(3, n), L%new_off(3, max_n1, n))
        allocate(L%old_e(5 * n), L%new_e(5 * n), L%u(NRAND, n), L%mvc(n), L%u5(5, n), L%acc_u(n), L%acc_pref(n))
        if (F%window) then
            allocate(L%w_live(n, 0:MGPU_FARM_DEPTH - 1), L%w_ia(n, 0:MGPU_FARM_DEPTH - 1), L%w_mv(n, 0:MGPU_FARM_DEPTH - 1), &
                     L%w_slot(n, 0:MGPU_FARM_DEPTH - 1), L%w_forced(n, 0:MGPU_FARM_DEPTH - 1), &
                     L%w_u(NRAND, n, 0:MGPU_FARM_DEPTH - 1), L%issued(n), L%resolved(n), L%pend_n(n), L%pend_forced(n), &
                     L%pend_u(NRAND, MGPU_FARM_DEPTH + 1, n), L%forced(n), L%verdict(n), L%slot_u(n))
            L%pend_n = 0; L%pend_forced = 0; L%issued = 0; L%resolved = 0
            L%w_head = 0; L%w_count = 0; L%outstanding = 0; L%in_flight = 0
        end if
        rc = mgpu_lane_site_buffer(F%engine, int(g, c_int), int(n, c_int), int(max_n1, c_int), staged)
        if (rc == MGPU_OK .and. c_associated(staged)) then
            call c_f_pointer(staged, L%sites, [3, max_n1, n])
        else
            allocate(L%sites_own(3, max_n1, n))
            L%sites => L%sites_own
        end if
        L%sites = 0.0_real64
        L%kind = MGPU_MOVE
    end subroutine alloc_lane

    subroutine mfarm_destroy() bind(C, name="mfarm_destroy")
        integer :: g
        if (allocated(F%res_type)) deallocate(F%res_type, F%n1, F%cap, F%first, F%cnt, F%fugacity)
        if (c_associated(F%mol_raw)) then
            call c_free(F%mol_raw)
            F%mol_raw = c_null_ptr
            nullify(F%mol)
        end if
        if (allocated(F%energy)) deallocate(F%energy)
        if (allocated(F%cxs)) deallocate(F%cxs)
        do g = 0, MGPU_LANES - 1
            if (allocated(F%lane(g)%rep)) then
                deallocate(F%lane(g)%rep, F%lane(g)%t, F%lane(g)%m, F%lane(g)%kind, F%lane(g)%accept, &
                           F%lane(g)%ia, F%lane(g)%move, F%lane(g)%cidx, F%lane(g)%sel_ia, F%lane(g)%sel_mv, &
                           F%lane(g)%sel_slot, F%lane(g)%new_com, &
                           F%lane(g)%new_off, F%lane(g)%old_e, F%lane(g)%new_e, F%lane(g)%u, F%lane(g)%mvc, F%lane(g)%u5, &
                           F%lane(g)%acc_u, F%lane(g)%acc_pref)
                if (allocated(F%lane(g)%sites_own)) deallocate(F%lane(g)%sites_own)
                if (allocated(F%lane(g)%w_live)) deallocate(F%lane(g)%w_live, F%lane(g)%w_ia, F%lane(g)%w_mv, F%lane(g)%w_slot, &
                    F%lane(g)%w_forced, F%lane(g)%w_u, F%lane(g)%issued, F%lane(g)%resolved, F%lane(g)%pend_n, &
                    F%lane(g)%pend_forced, F%lane(g)%pend_u, F%lane(g)%forced, F%lane(g)%verdict, F%lane(g)%slot_u)
                nullify(F%lane(g)%sites)
            end if
        end do
        F%ready = .false.
    end subroutine mfarm_destroy

    ! mirror record <- (com, offsets): explicit element copies, no array temporaries
    pure subroutine store_molecule(rec, com, off, n1)
        real(real64), intent(inout) :: rec(:)
        real(real64), intent(in) :: com(3), off(:, :)
        integer, intent(in) :: n1
        integer :: a, d
        do d = 1, 3
            rec(d) = com(d)
        end do
        do a = 1, n1
            do d = 1, 3
                rec(3 * a + d) = off(d, a)
            end do
        end do
    end subroutine store_molecule

    ! rotation by theta about Cartesian axis `axis`: RotationMatrix (src/helper_utils.f90:39-77)
    ! written out -- it mixes the two other components: X -> (Y, Z), Y -> (Z, X), Z -> (X, Y)
    pure subroutine rotate_offsets(off, n1, axis, theta)
        real(real64), intent(inout) :: off(:, :)
        integer, intent(in) :: n1, axis
        real(real64), intent(in) :: theta
        real(real64) :: c, sn, x, y
        integer :: p, q, a
        c = cos(theta)
        sn = sin(theta)
        p = mod(axis, 3) + 1
        q = mod(axis + 1, 3) + 1
        do a = 1, n1
            x = off(p, a)
            y = off(q, a)
            off(p, a) = c * x - sn * y
            off(q, a) = sn * x + c * y
        end do
    end subroutine rotate_offsets

    !---------------------------------------------------------------------------
    ! One trial move per replica of a lane, then queue its evaluation.
    !---------------------------------------------------------------------------
    function generate_and_submit(g) result(rc)
        integer, intent(in) :: g
        integer(c_int) :: rc
        integer :: i, j, k, r, ia, slot, n1, axis, d, a, mv, n
        integer(int64) :: c0, c1, c2, c3, cp1, cp2
        real(real64) :: x, draw, v(3), frac(3)
        type(lane_buffers), pointer :: L
        L => F%lane(g)
        rc = MGPU_OK
        L%nc = 0
        if (L%n == 0) return
        cp1 = 0; cp2 = 0
        call system_clock(c0)
        ! rng_kind 0: one serial draw from the intrinsic generator, the reference's stream
        if (F%rng_kind == 0) call random_number(L%u(:, 1:L%n))
        call system_clock(c3)
        ! One parallel region per lane and step (the fork / join is not free):
        !   phase 1  per chain: draw its numbers, select the move (monte_carlo.f90:50-75); selections that
        !            are no-ops in the reference (empty type, molecule_index = 0, rotation of an atom, full
        !            type) get slot 0;
        !   phase 2  (one thread) pack the chains that do attempt a move into candidates 1..nc;
        !   phase 3  per candidate: gather com / offsets from the host mirror -- one contiguous record, the
        !            random access is DRAM / TLB-latency bound and the threads overlap the misses -- and
        !            build the move.
        !$omp parallel num_threads(F%team) private(i, j, k, n1, d, x, axis, a, r, ia, slot, v, frac, n, mv, draw)
        ! the chains' own generators, blocks of RNG_BLOCK consecutive chains at a time (mgpu_rng_fill: four streams abreast)
        if (F%rng_kind /= 0) then
            !$omp do schedule(static)
            do i = 1, L%n, RNG_BLOCK
                k = mgpu_rng_fill(F%cxs(:, L%first + i), int(min(RNG_BLOCK, L%n - i + 1), c_int), int(NRAND, c_int), L%u(:, i))
            end do
            !$omp end do
        end if
        !$omp do schedule(static)
        do i = 1, L%n
            r = L%first + i
            ia = min(int(L%u(1, i) * F%n_active) + 1, F%n_active)      ! PickRandomResidueType
            n = F%cnt(ia, r)
            draw = L%u(3, i)
            if (draw <= F%p_translation) then
                mv = MV_TRANSLATION
            else if (draw <= F%p_rotation + F%p_translation) then
                mv = MV_ROTATION
            else if (L%u(10, i) <= PROB_CREATE_DELETE) then
                mv = MV_CREATION
            else
                mv = MV_DELETION
            end if
            if (.not. F%gcmc .and. mv > MV_ROTATION) mv = MV_ROTATION
            slot = 0
            if (mv == MV_ROTATION .and. F%n1(ia) == 1) then
                if (.not. F%gcmc) then
                    mv = MV_TRANSLATION                                 ! NVT farm of atoms: always translate
                    if (n > 0) slot = min(int(L%u(2, i) * n) + 1, n)
                end if                                                  ! GCMC: rotation.f90:45 returns
            else if (mv == MV_CREATION) then
                if (n < F%cap(ia)) slot = n + 1                         ! monte_carlo.f90:63; full: the reference aborts
            else
                if (n > 0) slot = min(int(L%u(2, i) * n) + 1, n)        ! PickRandomMoleculeIndex; 0: the drivers return
            end if
            L%sel_ia(i) = ia
            L%sel_mv(i) = mv
            L%sel_slot(i) = slot
            ! announce the mirror record phase 3 will gather: a prefetch does not stall this loop, so a thread has
            ! the DRAM misses of all its chains in flight instead of taking them one by one in phase 3
            if (slot > 0 .and. .not. F%device_build) then
                k = F%first(ia) + merge(1, slot, mv == MV_CREATION)
                call mgpu_host_prefetch(c_loc(F%mol(1, k, r)), int(8 * (3 + 3 * F%n1(ia)), c_int))
            end if
        end do
        !$omp end do
        !$omp master
        call system_clock(cp1)
        !$omp end master
        !$omp single
        j = 0
        do i = 1, L%n
            if (L%sel_slot(i) == 0) cycle
            j = j + 1
            L%cidx(j) = i
            L%ia(j) = L%sel_ia(i)
            L%move(j) = L%sel_mv(i)
            L%rep(j) = L%first + i - 1
            L%t(j) = F%res_type(L%sel_ia(i))
            L%m(j) = L%sel_slot(i) - 1
            select case (L%sel_mv(i))
            case (MV_CREATION)
                L%kind(j) = MGPU_CREATION
            case (MV_DELETION)
                L%kind(j) = MGPU_DELETION
            case default
                L%kind(j) = MGPU_MOVE
            end select
        end do
        L%nc = j
        L%skipped = L%skipped + (L%n - j)
        !$omp end single
        !$omp master
        call system_clock(cp2)
        !$omp end master
        !$omp do schedule(static)
        do j = 1, L%nc
            i = L%cidx(j)
            if (F%device_build) then
                ! the engine builds the geometry: hand over the move and the numbers its construction consumes
                ! (u4..u6 displacement / position, u7 angle, u8 axis -- the same draws the host construction below uses)
                L%mvc(j) = int(L%move(j), c_int)
                do d = 1, 5
                    L%u5(d, j) = L%u(3 + d, i)
                end do
                if (F%device_accept) then
                    ! what mc_acceptance_probability multiplies exp(-dE / T) by (monte_carlo_utils.f90:184-226), formed as
                    ! resolve_and_commit forms it, and the test's number
                    r = L%rep(j) + 1
                    ia = L%ia(j)
                    L%acc_u(j) = L%u(9, i)
                    select case (L%move(j))
                    case (MV_CREATION)
                        L%acc_pref(j) = F%fugacity(ia, r) * F%volume / real(F%cnt(ia, r) + 1, real64)
                    case (MV_DELETION)
                        L%acc_pref(j) = (real(F%cnt(ia, r) - 1, real64) + 1.0_real64) / (F%fugacity(ia, r) * F%volume)
                    case default
                        L%acc_pref(j) = 1.0_real64
                    end select
                end if
                cycle
            end if
            r = L%rep(j) + 1
            ia = L%ia(j)
            slot = L%m(j) + 1
            if (L%move(j) == MV_CREATION) slot = 1          ! geometry of molecule 1, create_molecule.f90:197-199
            n1 = F%n1(ia)
            ! explicit copies (an array-valued reshape costs a heap temporary per chain)
            k = F%first(ia) + slot
            do d = 1, 3
                L%new_com(d, j) = F%mol(d, k, r)
            end do
            do a = 1, n1
                do d = 1, 3
                    L%new_off(d, a, j) = F%mol(3 * a + d, k, r)
                end do
            end do
            select case (L%move(j))
            case (MV_TRANSLATION)
                ! translation.f90:104-110: rand_symmetric(3)*translation_step, then ApplyPBC
                ! (geometry_utils.f90:190: lo + modulo(pos - lo, L))
                if (.not. F%triclinic) then
                    do d = 1, 3
                        x = (L%new_com(d, j) + (L%u(3 + d, i) - 0.5_real64) * F%translation_step) - F%lo(d)
                        if (x < 0.0_real64 .or. x >= F%len(d)) x = modulo(x, F%len(d))
                        L%new_com(d, j) = F%lo(d) + x
                    end do
                else
                    ! geometry_utils.f90:196-211: s = H^-1 (r - r0), s <- modulo(s, 1), r <- r0 + H s
                    do d = 1, 3
                        v(d) = (L%new_com(d, j) + (L%u(3 + d, i) - 0.5_real64) * F%translation_step) - F%lo(d)
                    end do
                    do d = 1, 3
                        frac(d) = modulo(F%reciprocal(d, 1) * v(1) + F%reciprocal(d, 2) * v(2) + F%reciprocal(d, 3) * v(3), &
                                         1.0_real64)
                    end do
                    do d = 1, 3
                        L%new_com(d, j) = F%lo(d) + (F%matrix(d, 1) * frac(1) + F%matrix(d, 2) * frac(2) + &
                                                     F%matrix(d, 3) * frac(3))
                    end do
                end if
            case (MV_ROTATION)
                ! monte_carlo_utils.f90:54-64: theta = (u - 1/2)*rotation_step_angle, random Cartesian axis
                axis = int(L%u(8, i) * 3.0_real64) + 1
                call rotate_offsets(L%new_off(:, :, j), n1, axis, (L%u(7, i) - 0.5_real64) * F%rotation_step)
            case (MV_CREATION)
                ! create_molecule.f90:180-203: uniform position in the (orthorhombic) box, full rotation
                if (.not. F%triclinic) then
                    do d = 1, 3
                        L%new_com(d, j) = F%lo(d) + F%len(d) * L%u(3 + d, i)
                    end do
                else
                    ! create_molecule.f90:183-184: bounds(:, 1) + matmul(matrix, trial_pos)
                    do d = 1, 3
                        L%new_com(d, j) = F%lo(d) + (F%matrix(d, 1) * L%u(4, i) + F%matrix(d, 2) * L%u(5, i) + &
                                                     F%matrix(d, 3) * L%u(6, i))
                    end do
                end if
                if (n1 > 1) then
                    axis = int(L%u(8, i) * 3.0_real64) + 1
                    call rotate_offsets(L%new_off(:, :, j), n1, axis, L%u(7, i) * TWOPI)
                end if
            case (MV_DELETION)
                continue                                     ! the resident molecule is evaluated as it is
            end select
            do a = 1, n1
                L%sites(:, a, j) = L%new_com(:, j) + L%new_off(:, a, j)
            end do
        end do
        !$omp end do
        !$omp end parallel
        call system_clock(c1)
        if (L%nc > 0) then
            if (F%device_accept) then
                rc = mgpu_move_trial_decide_submit(F%engine, int(g, c_int), int(L%nc, c_int), L%rep, L%t, L%m, L%mvc, L%u5, &
                                                   F%translation_step, F%rotation_step, L%acc_u, L%acc_pref, F%temperature)
            else if (F%device_build) then
                rc = mgpu_move_trial_submit(F%engine, int(g, c_int), int(L%nc, c_int), L%rep, L%t, L%m, L%mvc, L%u5, &
                                            F%translation_step, F%rotation_step)
            else if (F%gcmc) then
                rc = mgpu_gcmc_trial_submit(F%engine, int(g, c_int), int(L%nc, c_int), L%rep, L%t, L%m, L%kind, &
                                            L%sites, int(F%max_n1, c_int))
            else
                rc = mgpu_trial_submit(F%engine, int(g, c_int), int(L%nc, c_int), L%rep, L%t, L%m, L%sites, &
                                       int(F%max_n1, c_int))
            end if
        end if
        call system_clock(c2)
        L%ticks(1) = L%ticks(1) + (c1 - c0)
        L%ticks(2) = L%ticks(2) + (c2 - c1)
        L%ticks(6) = L%ticks(6) + (cp1 - c0)                ! draw + move selection (phase 1, incl. the region's fork)
        L%ticks(7) = L%ticks(7) + (c1 - cp2)                ! mirror gather + move construction (phase 3, incl. the join)
    end function generate_and_submit

    !---------------------------------------------------------------------------
    ! Collect a lane's energies, apply the acceptance rule per replica
    ! (mc_acceptance_probability, monte_carlo_utils.f90:184-226), update the host mirrors and
    ! running energies (AcceptMove / AcceptCreationMove / AcceptDeletionMove) and queue the commit.
    !---------------------------------------------------------------------------
    function resolve_and_commit(g) result(rc)
        integer, intent(in) :: g
        integer(c_int) :: rc
        integer :: i, j, k, r, ia, slot, n1, ne, last, base, o
        integer(int64) :: c0, c1, c2, c3
        integer(int64) :: k_tt, k_t, k_rt, k_r, k_ct, k_c, k_dt, k_d
        real(real64) :: delta_e, probability, nn, phi, e_old, e_new
        logical :: yes
        type(lane_buffers), pointer :: L
        L => F%lane(g)
        rc = MGPU_OK
        if (L%nc == 0) return
        call system_clock(c0)
        if (F%device_accept) then
            ne = 5
            rc = mgpu_trial_decide_wait(F%engine, int(g, c_int), L%old_e, L%new_e, L%accept)   ! rows of 5 + the outcomes
        else if (F%gcmc .or. F%device_build) then
            ne = 5
            rc = mgpu_gcmc_trial_wait(F%engine, int(g, c_int), L%old_e, L%new_e)   ! rows of 5
        else
            ne = 3
            rc = mgpu_trial_wait(F%engine, int(g, c_int), L%old_e, L%new_e)        ! rows of 3
        end if
        if (rc /= MGPU_OK) return
        call system_clock(c1)
        k_tt = 0; k_t = 0; k_rt = 0; k_r = 0; k_ct = 0; k_c = 0; k_dt = 0; k_d = 0
        !$omp parallel do num_threads(F%team) schedule(static) &
        !$omp& private(i, k, r, ia, slot, n1, delta_e, probability, nn, phi, last, base, o, e_old, e_new, yes) &
        !$omp& reduction(+:k_tt, k_t, k_rt, k_r, k_ct, k_c, k_dt, k_d)
        do j = 1, L%nc
            i = L%cidx(j)
            r = L%rep(j) + 1
            ia = L%ia(j)
            n1 = F%n1(ia)
            base = F%first(ia)
            o = ne * (j - 1)
            e_old = 0.0_real64
            e_new = 0.0_real64
            do k = 1, ne                                                   ! old%total, new%total
                e_old = e_old + L%old_e(o + k)
                e_new = e_new + L%new_e(o + k)
            end do
            delta_e = e_new - e_old
            if (F%device_accept) then
                ! the engine applied the rule (same totals, same prefactor, same number) and committed: L%accept holds the outcome
                select case (L%move(j))
                case (MV_CREATION)
                    k_ct = k_ct + 1
                case (MV_DELETION)
                    k_dt = k_dt + 1
                case (MV_TRANSLATION)
                    k_tt = k_tt + 1
                case default
                    k_rt = k_rt + 1
                end select
                yes = L%accept(j) /= 0
            else
            select case (L%move(j))
            case (MV_CREATION)
                nn = real(F%cnt(ia, r) + 1, real64)                        ! N already incremented, create_molecule.f90:64
                phi = F%fugacity(ia, r)
                probability = min(1.0_real64, (phi * F%volume / nn) * exp(-delta_e / F%temperature))
                k_ct = k_ct + 1
            case (MV_DELETION)
                nn = real(F%cnt(ia, r) - 1, real64)                        ! N already decremented, delete_molecule.f90:73
                phi = F%fugacity(ia, r)
                probability = min(1.0_real64, ((nn + 1.0_real64) / (phi * F%volume)) * exp(-delta_e / F%temperature))
                k_dt = k_dt + 1
            case (MV_TRANSLATION)
                probability = min(1.0_real64, exp(-delta_e / F%temperature))
                k_tt = k_tt + 1
            case default
                probability = min(1.0_real64, exp(-delta_e / F%temperature))
                k_rt = k_rt + 1
            end select
            yes = L%u(9, i) <= probability
            end if
            if (yes) then
                L%accept(j) = 1
                slot = L%m(j) + 1
                select case (L%move(j))
                case (MV_CREATION)
                    if (.not. F%device_build) call store_molecule(F%mol(:, base + slot, r), L%new_com(:, j), L%new_off(:, :, j), n1)
                    F%cnt(ia, r) = F%cnt(ia, r) + 1
                    k_c = k_c + 1
                case (MV_DELETION)
                    last = F%cnt(ia, r)                                    ! RemoveMolecule, delete_molecule.f90:107-114
                    if (.not. F%device_build) then
                        do k = 1, 3 + 3 * n1
                            F%mol(k, base + slot, r) = F%mol(k, base + last, r)
                        end do
                    end if
                    F%cnt(ia, r) = last - 1
                    k_d = k_d + 1
                case default
                    if (.not. F%device_build) call store_molecule(F%mol(:, base + slot, r), L%new_com(:, j), L%new_off(:, :, j), n1)
                    if (L%move(j) == MV_TRANSLATION) then
                        k_t = k_t + 1
                    else
                        k_r = k_r + 1
                    end if
                end select
                ! monte_carlo_utils.f90:416-419, create_molecule.f90:107-112, delete_molecule.f90:137-142
                do k = 1, ne
                    F%energy(k, r) = F%energy(k, r) + L%new_e(o + k) - L%old_e(o + k)
                end do
            else
                L%accept(j) = 0
            end if
        end do
        !$omp end parallel do
        L%counters = L%counters + [k_tt, k_t, k_rt, k_r, k_ct, k_c, k_dt, k_d]
        L%accepted = L%accepted + k_t + k_r + k_c + k_d
        L%trials = L%trials + L%nc
        call system_clock(c2)
        if (.not. F%device_accept) then
            rc = mgpu_commit_submit(F%engine, int(g, c_int), int(L%nc, c_int), L%rep, L%t, L%m, L%kind, &
                                    c_null_ptr, int(F%max_n1, c_int), L%accept)
        end if
        call system_clock(c3)
        L%nc = 0
        L%ticks(3) = L%ticks(3) + (c1 - c0)
        L%ticks(4) = L%ticks(4) + (c2 - c1)
        L%ticks(5) = L%ticks(5) + (c3 - c2)
    end function resolve_and_commit

    !---------------------------------------------------------------------------
    ! Window mode.  Move selection for chain r from its draws u and its counts: monte_carlo.f90:50-75, exactly
    ! generate_and_submit's phase 1 (slot 0 = the selections that are no-ops in the reference).
    !---------------------------------------------------------------------------
    subroutine select_move(r, u, ia, mv, slot)
        integer, intent(in) :: r
        real(real64), intent(in) :: u(NRAND)
        integer, intent(out) :: ia, mv, slot
        integer :: n
        real(real64) :: draw
        ia = min(int(u(1) * F%n_active) + 1, F%n_active)              ! PickRandomResidueType
        n = F%cnt(ia, r)
        draw = u(3)
        if (draw <= F%p_translation) then
            mv = MV_TRANSLATION
        else if (draw <= F%p_rotation + F%p_translation) then
            mv = MV_ROTATION
        else if (u(10) <= PROB_CREATE_DELETE) then
            mv = MV_CREATION
        else
            mv = MV_DELETION
        end if
        if (.not. F%gcmc .and. mv > MV_ROTATION) mv = MV_ROTATION
        slot = 0
        if (mv == MV_ROTATION .and. F%n1(ia) == 1) then
            if (.not. F%gcmc) then
                mv = MV_TRANSLATION                                     ! NVT farm of atoms: always translate
                if (n > 0) slot = min(int(u(2) * n) + 1, n)
            end if                                                      ! GCMC: rotation.f90:45 returns
        else if (mv == MV_CREATION) then
            if (n < F%cap(ia)) slot = n + 1                             ! monte_carlo.f90:63; full: the reference aborts
        else
            if (n > 0) slot = min(int(u(2) * n) + 1, n)                 ! PickRandomMoleculeIndex; 0: the drivers return
        end if
    end subroutine select_move

    ! what mc_acceptance_probability multiplies exp(-dE / T) by (monte_carlo_utils.f90:184-226), as resolve_and_commit forms it
    pure function acceptance_prefactor(mv, ia, r) result(pref)
        integer, intent(in) :: mv, ia, r
        real(real64) :: pref
        select case (mv)
        case (MV_CREATION)
            pref = F%fugacity(ia, r) * F%volume / real(F%cnt(ia, r) + 1, real64)
        case (MV_DELETION)
            pref = (real(F%cnt(ia, r) - 1, real64) + 1.0_real64) / (F%fugacity(ia, r) * F%volume)
        case default
            pref = 1.0_real64
        end select
    end function acceptance_prefactor

    ! Queue the lane's next window: one record per chain -- a step waiting to be sent again, else the chain's next step
    ! (fresh draws, in the chain's own order), else a filler.
    function issue_window(g, n_target) result(rc)
        integer, intent(in) :: g, n_target
        integer(c_int) :: rc
        integer :: i, b, r, ia, mv, slot, k, d
        integer(int64) :: c0, c1, c2
        logical :: filled
        type(lane_buffers), pointer :: L
        L => F%lane(g)
        rc = MGPU_OK
        call system_clock(c0)
        b = mod(L%w_head + L%w_count, MGPU_FARM_DEPTH)
        ! rng_kind 0: the intrinsic generator, serially, for the chains that take a fresh step
        if (F%rng_kind == 0) then
            do i = 1, L%n
                if (L%pend_n(i) == 0 .and. L%issued(i) < n_target) call random_number(L%w_u(:, i, b))
            end do
        end if
        ! the usual case -- every chain takes a fresh step -- draws in blocks (mgpu_rng_fill: four streams abreast)
        filled = F%rng_kind /= 0 .and. all(L%pend_n(1:L%n) == 0) .and. all(L%issued(1:L%n) < n_target)
        if (filled) then
            do i = 1, L%n, RNG_BLOCK
                k = mgpu_rng_fill(F%cxs(:, L%first + i), int(min(RNG_BLOCK, L%n - i + 1), c_int), int(NRAND, c_int), L%w_u(:, i, b))
            end do
        end if
        ! (one thread: a chain's record costs ~25 ns and a lane holds at most a few thousand chains -- measured round 5, a team of two made
        !  this loop 31 us at 512 chains against 12: the fork / join of a region outweighs a loop this short)
        do i = 1, L%n
            r = L%first + i
            L%rep(i) = int(r - 1, c_int)
            L%forced(i) = 0
            if (L%pend_n(i) > 0) then
                L%w_u(:, i, b) = L%pend_u(:, 1, i)
                do k = 2, L%pend_n(i)
                    L%pend_u(:, k - 1, i) = L%pend_u(:, k, i)
                end do
                L%pend_n(i) = L%pend_n(i) - 1
                L%forced(i) = int(L%pend_forced(i), c_int)
                L%pend_forced(i) = 0
            else if (L%issued(i) < n_target) then
                if (F%rng_kind /= 0 .and. .not. filled) call chain_random(r, L%w_u(:, i, b))
                L%issued(i) = L%issued(i) + 1
            else
                L%w_live(i, b) = 0                                      ! nothing left for this chain: a filler
                L%t(i) = 0; L%m(i) = 0; L%mvc(i) = 0
                L%acc_u(i) = 0.0_real64; L%acc_pref(i) = 1.0_real64
                L%u5(:, i) = 0.0_real64
                L%slot_u(i) = 0.0_real64
                cycle
            end if
            call select_move(r, L%w_u(:, i, b), ia, mv, slot)
            L%w_live(i, b) = 1
            L%w_ia(i, b) = ia; L%w_mv(i, b) = mv; L%w_slot(i, b) = slot; L%w_forced(i, b) = L%forced(i)
            L%t(i) = int(F%res_type(ia), c_int)
            do d = 1, 5
                L%u5(d, i) = L%w_u(3 + d, i, b)                         ! u4..u6 displacement / position, u7 angle, u8 axis
            end do
            L%acc_u(i) = L%w_u(9, i, b)
            if (F%gcmc) then
                ! Insertion / deletion farms: the residue type and the kind of move do not depend on the molecule count, the
                ! slot and the prefactor do -- and the count depends on the outcome of a window that may still be in flight.
                ! The engine completes the record from the replica's count when the launch runs (slot_u, phi V); the slot
                ! select_move has just computed from THIS side's count is not used -- resolve_window replays the selection
                ! once the earlier windows have been followed.  The one no-op that does not depend on the count: the
                ! rotation of an atom (rotation.f90:45).
                L%m(i) = 0
                L%mvc(i) = int(merge(0, mv, mv == MV_ROTATION .and. F%n1(ia) == 1), c_int)
                L%slot_u(i) = L%w_u(2, i, b)
                L%acc_pref(i) = merge(F%fugacity(ia, r) * F%volume, 1.0_real64, mv >= MV_CREATION)
            else
                L%m(i) = int(max(slot, 1) - 1, c_int)
                L%mvc(i) = int(merge(mv, 0, slot > 0), c_int)           ! a no-op selection: the engine does nothing for it
                L%acc_pref(i) = acceptance_prefactor(mv, ia, r)
            end if
        end do
        call system_clock(c1)
        if (F%gcmc) then
            rc = mgpu_farm_window_submit(F%engine, int(g, c_int), int(L%n, c_int), L%rep, L%t, L%m, L%mvc, L%forced, L%u5, &
                                         L%acc_u, L%acc_pref, c_loc(L%slot_u), F%translation_step, F%rotation_step, F%temperature)
        else
            rc = mgpu_farm_window_submit(F%engine, int(g, c_int), int(L%n, c_int), L%rep, L%t, L%m, L%mvc, L%forced, L%u5, &
                                         L%acc_u, L%acc_pref, c_null_ptr, F%translation_step, F%rotation_step, F%temperature)
        end if
        call system_clock(c2)
        L%in_flight = L%in_flight + count(L%w_live(1:L%n, b) == 1)
        L%w_count = L%w_count + 1
        L%ticks(1) = L%ticks(1) + (c1 - c0)
        L%ticks(2) = L%ticks(2) + (c2 - c1)
    end function issue_window

    ! Collect the lane's oldest window and follow its outcomes: every verdict the device took is checked against the
    ! driver's own rule (mc_acceptance_probability, monte_carlo_utils.f90:184-226); what the device left undecided the
    ! driver decides and sends again.
    function resolve_window(g) result(rc)
        integer, intent(in) :: g
        integer(c_int) :: rc
        integer :: i, b, r, ia, mv, slot, k, o, v
        integer(int64) :: c0, c1, c2
        integer(int64) :: k_tt, k_t, k_rt, k_r, k_ct, k_c, k_dt, k_d, n_skip, n_done, n_und, n_back
        real(real64) :: e_old, e_new, delta_e, probability
        logical :: yes, bad
        type(lane_buffers), pointer :: L
        L => F%lane(g)
        rc = MGPU_OK
        call system_clock(c0)
        b = L%w_head
        rc = mgpu_farm_window_wait(F%engine, int(g, c_int), L%old_e, L%new_e, L%verdict)
        if (rc /= MGPU_OK) return
        call system_clock(c1)
        k_tt = 0; k_t = 0; k_rt = 0; k_r = 0; k_ct = 0; k_c = 0; k_dt = 0; k_d = 0; n_skip = 0; n_done = 0; n_und = 0
        n_back = 0
        bad = .false.
        ! (one thread, as in issue_window: a chain costs an exp and a few adds)
        do i = 1, L%n
            if (L%w_live(i, b) == 0) cycle
            n_back = n_back + 1
            r = L%first + i
            ia = L%w_ia(i, b); mv = L%w_mv(i, b); slot = L%w_slot(i, b)
            v = L%verdict(i)
            if (F%gcmc .and. v /= 4) then
                ! the selection again, now that every earlier window of the chain has been followed: this side's count is the
                ! count the launch saw (a record that did nothing because its replica waits -- verdict 4 -- is sent again as it is)
                call select_move(r, L%w_u(:, i, b), ia, mv, slot)
                if ((slot == 0) .neqv. (v == 5)) bad = .true.           ! both sides must find the same steps empty
            end if
            if (v == 4) then
                ! the window ran behind an undecided step of this chain and did nothing: send the step again, in order
                L%pend_n(i) = L%pend_n(i) + 1
                L%pend_u(:, L%pend_n(i), i) = L%w_u(:, i, b)
                cycle
            end if
            if (slot == 0) then
                ! a no-op selection (empty type, full type, rotation of an atom): the step is spent
                n_skip = n_skip + 1
                n_done = n_done + 1
                L%resolved(i) = L%resolved(i) + 1
                cycle
            end if
            o = 5 * (i - 1)
            e_old = 0.0_real64
            e_new = 0.0_real64
            do k = 1, 5                                                   ! old%total, new%total
                e_old = e_old + L%old_e(o + k)
                e_new = e_new + L%new_e(o + k)
            end do
            delta_e = e_new - e_old
            probability = min(1.0_real64, acceptance_prefactor(mv, ia, r) * exp(-delta_e / F%temperature))
            yes = L%w_u(9, i, b) <= probability
            if (v == 2) then
                ! left to the driver: its decision goes back with the step (in front of anything already waiting)
                do k = L%pend_n(i), 1, -1
                    L%pend_u(:, k + 1, i) = L%pend_u(:, k, i)
                end do
                L%pend_u(:, 1, i) = L%w_u(:, i, b)
                L%pend_n(i) = L%pend_n(i) + 1
                L%pend_forced(i) = merge(1, 2, yes)
                n_und = n_und + 1
                cycle
            end if
            if (L%w_forced(i, b) /= 0) then
                yes = L%w_forced(i, b) == 1                               ! the driver's earlier decision, obeyed
                if ((v == 1) .neqv. yes) bad = .true.
            else if ((v == 1) .neqv. yes) then
                bad = .true.                                              ! outside the margin the two rules cannot differ
            end if
            select case (mv)
            case (MV_CREATION)
                k_ct = k_ct + 1
            case (MV_DELETION)
                k_dt = k_dt + 1
            case (MV_TRANSLATION)
                k_tt = k_tt + 1
            case default
                k_rt = k_rt + 1
            end select
            if (v == 1) then
                select case (mv)
                case (MV_CREATION)
                    F%cnt(ia, r) = F%cnt(ia, r) + 1
                    k_c = k_c + 1
                case (MV_DELETION)
                    F%cnt(ia, r) = F%cnt(ia, r) - 1
                    k_d = k_d + 1
                case (MV_TRANSLATION)
                    k_t = k_t + 1
                case default
                    k_r = k_r + 1
                end select
                ! monte_carlo_utils.f90:416-419, create_molecule.f90:107-112, delete_molecule.f90:137-142
                do k = 1, 5
                    F%energy(k, r) = F%energy(k, r) + L%new_e(o + k) - L%old_e(o + k)
                end do
            end if
            n_done = n_done + 1
            L%resolved(i) = L%resolved(i) + 1
        end do
        L%in_flight = L%in_flight - n_back
        L%counters = L%counters + [k_tt, k_t, k_rt, k_r, k_ct, k_c, k_dt, k_d]
        L%accepted = L%accepted + k_t + k_r + k_c + k_d
        L%trials = L%trials + k_tt + k_rt + k_ct + k_dt
        L%skipped = L%skipped + n_skip
        L%outstanding = L%outstanding - n_done
        L%undecided = L%undecided + n_und
        L%w_head = mod(L%w_head + 1, MGPU_FARM_DEPTH)
        L%w_count = L%w_count - 1
        call system_clock(c2)
        L%ticks(3) = L%ticks(3) + (c1 - c0)
        L%ticks(4) = L%ticks(4) + (c2 - c1)
        if (bad) rc = 7                   ! a device decision the driver's rule contradicts: never outside the margin
    end function resolve_window

    ! n_steps move selections of every chain through windows: a driver's lanes take turns; a lane keeps up to `depth` windows
    ! in flight while it has steps to send and collects its oldest one each turn.  n_drv > 1: n_drv host threads, driver d
    ! with the lanes d, d + n_drv, ... (the lanes share nothing on either side of the C ABI: own stream, own ring of
    ! windows, own chains).
    function run_windows(n_steps, n_drv) result(rc)
        integer, intent(in) :: n_steps, n_drv
        integer(c_int) :: rc
        integer :: g, d
        integer(c_int) :: rc_drv(0:MGPU_LANES - 1)
        type(lane_buffers), pointer :: L
        rc = MGPU_OK
        do g = 0, F%n_lanes - 1
            L => F%lane(g)
            if (L%n == 0) cycle
            L%issued = 0; L%resolved = 0
            L%outstanding = int(L%n, int64) * int(n_steps, int64)
            L%in_flight = 0
        end do
        rc_drv = MGPU_OK
        if (n_drv <= 1) then
            rc_drv(0) = drive_windows(0, 1, n_steps)
        else
            !$omp parallel num_threads(n_drv) private(d) proc_bind(spread)
            d = omp_get_thread_num()
            rc_drv(d) = drive_windows(d, n_drv, n_steps)
            !$omp end parallel
        end if
        do d = 0, MGPU_LANES - 1
            if (rc_drv(d) /= MGPU_OK) rc = rc_drv(d)
        end do
        do g = 0, F%n_lanes - 1
            F%undecided = F%undecided + F%lane(g)%undecided
            F%lane(g)%undecided = 0
        end do
    end function run_windows

    function drive_windows(d, n_drv, n_steps) result(rc)
        integer, intent(in) :: d, n_drv, n_steps
        integer(c_int) :: rc
        integer :: g, depth
        logical :: busy
        type(lane_buffers), pointer :: L
        rc = MGPU_OK
        depth = max(1, min(F%depth, int(MGPU_FARM_DEPTH)))
        do
            busy = .false.
            do g = d, F%n_lanes - 1, n_drv
                L => F%lane(g)
                if (L%n == 0) cycle
                do while (L%w_count < depth .and. L%outstanding - L%in_flight > 0)
                    rc = issue_window(g, n_steps)
                    if (rc /= MGPU_OK) return
                end do
                if (L%w_count > 0) then
                    rc = resolve_window(g)
                    if (rc /= MGPU_OK) return
                end if
                if (L%outstanding > 0) busy = .true.
            end do
            if (.not. busy) exit
        end do
    end function drive_windows

    !---------------------------------------------------------------------------
    ! Advance every chain by n_steps move selections.  out = trials, accepted, skipped selections.
    !---------------------------------------------------------------------------
    function mfarm_run(n_steps, out) bind(C, name="mfarm_run") result(rc)
        integer(c_int), value :: n_steps
        real(c_double), intent(out) :: out(3)
        integer(c_int) :: rc
        integer :: step, g, lu, ios, nlen, n_drv, d
        integer(c_int) :: rc_lane(0:MGPU_LANES - 1)
        integer :: elen, eios
        character(len=16) :: envval
        integer(int64), allocatable :: slog(:, :)
        integer(int64) :: rate
        character(len=512) :: logpath
        rc = MGPU_OK
        out = 0.0_real64
        if (.not. F%ready) then
            rc = 5
            return
        end if
        ! diagnostic: MFARM_STEPLOG=<file> appends the host timers (microseconds) after every step
        call get_environment_variable("MFARM_STEPLOG", logpath, nlen, ios)
        if (ios == 0 .and. nlen > 0 .and. n_steps > 0) allocate(slog(7, 0:n_steps))
        ! Two ways to drive the lanes.  (a) lock step on the calling thread (default): the lanes take turns, which also
        ! staggers their kernels on the device -- the persistent pair sweeps of two lanes then run one after the other.
        ! (b) MFARM_LANE_THREADS=1: one host thread per lane, each with its own OpenMP team, runs its lane's resolve ->
        ! commit -> generate -> submit loop on its own (the engine's lanes are independent: own stream, scratch, staging,
        ! profiling slots).  Measured on MI355X at the default bench size: the host stops being the limit (it waits
        ! 60 % of the time) but the free-running lanes' pair sweeps share the device and stretch (100 -> 104-112 us),
        ! 5.3 M against 5.9 M accepted moves/s in lock step -- kept as an option for hosts too slow to keep up.
        ! MFARM_LANE_THREADS=<d>: d driver threads (1 = as many as lanes, the round-2 meaning), each running the lock-step
        ! loop over ITS lanes (d, d + n_drv, ...) with a team of n_threads / n_drv -- a driver thread with two lanes
        ! overlaps its own host work on one lane with the GPU's work on the other, and the driver threads overlap
        ! each other's host work (the grand-canonical boxes are host-bound with one driver)
        F%lane_threads = .false.
        n_drv = F%n_drivers
        call get_environment_variable("MFARM_LANE_THREADS", envval, elen, eios)
        if (eios == 0 .and. elen > 0) then
            read(envval(1:elen), *, iostat=eios) n_drv
            if (eios /= 0) n_drv = 1
            if (n_drv == 1) n_drv = F%n_lanes
        end if
        n_drv = max(1, min(n_drv, F%n_lanes))
        F%lane_threads = n_drv > 1 .and. F%rng_kind /= 0 .and. .not. allocated(slog) .and. F%n_threads >= 2 * n_drv
        if (n_steps > 0 .and. F%window) then
            ! (the window loops run on one thread per driver: no teams)
            if (.not. (F%rng_kind /= 0 .and. F%n_threads >= n_drv)) n_drv = 1
            F%lane_threads = .false.
            F%team = max(1, F%n_threads / n_drv)
            rc = mgpu_set_host_team(F%engine, int(F%team, c_int))
            if (rc /= MGPU_OK) return
            rc = run_windows(int(n_steps), n_drv)
            if (rc /= MGPU_OK) return
        else if (n_steps > 0 .and. F%lane_threads) then
            F%team = max(1, F%n_threads / n_drv)
            rc = mgpu_set_host_team(F%engine, int(F%team, c_int))
            if (rc /= MGPU_OK) return
            call omp_set_max_active_levels(2)
            rc_lane = MGPU_OK
            !$omp parallel num_threads(n_drv) private(d, g, step) proc_bind(spread)
            d = omp_get_thread_num()
            do g = d, F%n_lanes - 1, n_drv
                rc_lane(g) = generate_and_submit(g)
            end do
            do step = 1, n_steps
                do g = d, F%n_lanes - 1, n_drv
                    if (rc_lane(g) /= MGPU_OK) cycle
                    rc_lane(g) = resolve_and_commit(g)
                    if (rc_lane(g) == MGPU_OK .and. step < n_steps) rc_lane(g) = generate_and_submit(g)
                end do
            end do
            !$omp end parallel
            do g = 0, F%n_lanes - 1
                if (rc_lane(g) /= MGPU_OK) rc = rc_lane(g)
            end do
            if (rc == MGPU_OK) rc = mgpu_synchronize(F%engine)
        else if (n_steps > 0) then
            F%team = F%n_threads
            rc = mgpu_set_host_team(F%engine, int(F%team, c_int))
            if (rc /= MGPU_OK) return
            do g = 0, F%n_lanes - 1
                rc = generate_and_submit(g)
                if (rc /= MGPU_OK) return
            end do
            if (allocated(slog)) call lane_ticks(slog(:, 0))
            do step = 1, n_steps
                do g = 0, F%n_lanes - 1
                    rc = resolve_and_commit(g)
                    if (rc /= MGPU_OK) return
                    if (step < n_steps) then
                        rc = generate_and_submit(g)
                        if (rc /= MGPU_OK) return
                    end if
                end do
                if (allocated(slog)) call lane_ticks(slog(:, step))
            end do
            rc = mgpu_synchronize(F%engine)
        end if
        ! fold the lanes' accumulators into the farm's totals
        do g = 0, F%n_lanes - 1
            F%trials = F%trials + F%lane(g)%trials
            F%accepted = F%accepted + F%lane(g)%accepted
            F%skipped = F%skipped + F%lane(g)%skipped
            F%counters = F%counters + F%lane(g)%counters
            F%ticks = F%ticks + F%lane(g)%ticks
            F%lane(g)%trials = 0; F%lane(g)%accepted = 0; F%lane(g)%skipped = 0
            F%lane(g)%counters = 0; F%lane(g)%ticks = 0
        end do
        if (allocated(slog)) then
            call system_clock(count_rate=rate)
            open(newunit=lu, file=logpath(1:nlen), position="append", action="write", iostat=ios)
            if (ios == 0) then
                write(lu, '(a, i0)') "# mfarm_run steps=", n_steps
                do step = 1, n_steps
                    write(lu, '(i6, 7f10.1)') step, real(slog(:, step) - slog(:, step - 1), real64) * 1.0d6 / real(rate, real64)
                end do
                close(lu)
            end if
        end if
        out(1) = real(F%trials, real64)
        out(2) = real(F%accepted, real64)
        out(3) = real(F%skipped, real64)
    end function mfarm_run

    ! host timers so far: the farm's totals plus what the lanes have accumulated since the last fold
    subroutine lane_ticks(t)
        integer(int64), intent(out) :: t(7)
        integer :: g
        t = F%ticks
        do g = 0, F%n_lanes - 1
            t = t + F%lane(g)%ticks
        end do
    end subroutine lane_ticks

    ! Select the farm slot (0 .. MAX_FARMS - 1) the following calls act on; returns 0, or 1 for a slot out of range.
    function mfarm_select(slot) bind(C, name="mfarm_select") result(rc)
        integer(c_int), value :: slot
        integer(c_int) :: rc
        rc = 1
        if (slot < 0 .or. slot >= MAX_FARMS) return
        F => farms(slot)
        rc = 0
    end function mfarm_select

    ! device_build /= 0: the NEXT farm created builds its trial moves on the device from the frames the caller uploaded with
    ! mgpu_replica_set_frames (orthorhombic boxes); 0: on the host from its mirror (the default); 2: the engine also applies
    ! the acceptance rule and commits accepted candidates itself (mgpu_move_trial_decide_submit) -- the farm then only
    ! draws the numbers, selects the moves and keeps its counts, energies and counters in step with the outcomes
    subroutine mfarm_configure(device_build) bind(C, name="mfarm_configure")
        integer(c_int), value :: device_build
        want_device_build = device_build /= 0
        want_device_accept = device_build == 2               ! 2: the engine also decides and commits
        want_window = device_build == 3                      ! 3: one launch per lane step (mgpu_farm_window_submit)
    end subroutine mfarm_configure

    ! window mode of the selected farm: out(1) = 1 if its steps go through mgpu_farm_window_submit, out(2) = windows of a
    ! lane in flight, out(3) = steps the device left to the driver so far
    subroutine mfarm_window_mode(out) bind(C, name="mfarm_window_mode")
        real(c_double), intent(out) :: out(3)
        out(1) = merge(1.0_real64, 0.0_real64, F%window)
        out(2) = real(F%depth, real64)
        out(3) = real(F%undecided, real64)
    end subroutine mfarm_window_mode

    subroutine mfarm_set_window_depth(depth) bind(C, name="mfarm_set_window_depth")
        integer(c_int), value :: depth
        F%depth = max(1, min(int(depth), int(MGPU_FARM_DEPTH)))
    end subroutine mfarm_set_window_depth

    ! Driver threads of mfarm_run (1: the calling thread drives all lanes in lock step; d > 1: d threads, each driving
    ! lanes d0, d0 + d, ... with a team of n_threads / d).  Measured on MI355X (round 3): two drivers on four lanes lift the
    ! host-bound grand-canonical farms (CO2 box 14.2 -> 19.8 M, framework + water 4.6 -> 5.3 M accepted moves/s) and
    ! change nothing for the GPU-bound 10 125-atom box (7.16 M either way).
    subroutine mfarm_set_drivers(n) bind(C, name="mfarm_set_drivers")
        integer(c_int), value :: n
        F%n_drivers = max(1, int(n))
    end subroutine mfarm_set_drivers

    ! trial / accepted counts: translations, rotations, creations, deletions (counter_type,
    ! src/simulation_state.f90:19-31)
    subroutine mfarm_get_counters(c) bind(C, name="mfarm_get_counters")
        real(c_double), intent(out) :: c(8)
        c = real(F%counters, real64)
    end subroutine mfarm_get_counters

    ! AdjustMoveStepSizes (src/monte_carlo_utils.f90:99-130), as written in the reference
    ! (including its min(..*1.95, MIN_ROTATION_ANGLE) branch), on the farm-wide counters.
    subroutine mfarm_recalibrate(steps) bind(C, name="mfarm_recalibrate")
        real(c_double), intent(out) :: steps(2)
        real(real64) :: acc
        if (F%counters(1) > MIN_TRIALS_FOR_RECALIBRATION) then
            acc = real(F%counters(2), real64) / real(F%counters(1), real64)
            if (acc - TARGET_ACCEPTANCE > TOL_ACCEPTANCE) then
                F%translation_step = min(F%translation_step * 1.05d0, MAX_TRANSLATION_STEP)
            else if (acc - TARGET_ACCEPTANCE < TOL_ACCEPTANCE) then
                F%translation_step = max(F%translation_step * 0.95d0, MIN_TRANSLATION_STEP)
            end if
        end if
        if (F%counters(3) > MIN_TRIALS_FOR_RECALIBRATION) then
            acc = real(F%counters(4), real64) / real(F%counters(3), real64)
            if (acc - TARGET_ACCEPTANCE > TOL_ACCEPTANCE) then
                F%rotation_step = min(F%rotation_step * 1.05d0, MAX_ROTATION_ANGLE)
            else if (acc - TARGET_ACCEPTANCE < TOL_ACCEPTANCE) then
                F%rotation_step = min(F%rotation_step * 1.95d0, MIN_ROTATION_ANGLE)
            end if
        end if
        steps(1) = F%translation_step
        steps(2) = F%rotation_step
    end subroutine mfarm_recalibrate

    ! Test hook: seed n_chains generators as mfarm_create would and return the first n_per numbers of each stream,
    ! u(n_per, n_chains).  A live farm's chain generators are set aside and put back, and the intrinsic generator
    ! (rng_kind 0, the reference's stream) is not reseeded.
    subroutine mfarm_rng_sample(seed, n_chains, n_per, u) bind(C, name="mfarm_rng_sample")
        integer(c_int), value :: seed, n_chains, n_per
        real(c_double), intent(out) :: u(n_per, n_chains)
        integer(int64), allocatable :: keep(:, :)
        real(real64) :: v(NRAND)
        integer :: r, k, got
        if (allocated(F%cxs)) call move_alloc(F%cxs, keep)
        call seed_chain_streams(int(seed), int(n_chains))
        do r = 1, n_chains
            got = 0
            do while (got < n_per)
                call chain_random(r, v)
                k = min(NRAND, n_per - got)
                u(got + 1:got + k, r) = v(1:k)
                got = got + k
            end do
        end do
        deallocate(F%cxs)
        if (allocated(keep)) call move_alloc(keep, F%cxs)
    end subroutine mfarm_rng_sample

    ! host wall time spent in: trial generation, trial submit, waiting for the GPU, Metropolis
    ! resolution, commit submit, and inside generation: random numbers, mirror gathers (seconds)
    subroutine mfarm_get_timers(t) bind(C, name="mfarm_get_timers")
        real(c_double), intent(out) :: t(7)
        integer(int64) :: rate
        call system_clock(count_rate=rate)
        t = real(F%ticks, real64) / real(rate, real64)
    end subroutine mfarm_get_timers

    ! running energies (non_coulomb, coulomb, recip_coulomb, ewald_self, intra_coulomb) of one replica
    subroutine mfarm_get_energy(replica, e) bind(C, name="mfarm_get_energy")
        integer(c_int), value :: replica
        real(c_double), intent(out) :: e(5)
        e = F%energy(:, replica + 1)
    end subroutine mfarm_get_energy

    ! current molecule counts, (n_active, R) column-major
    subroutine mfarm_get_counts(cnt) bind(C, name="mfarm_get_counts")
        integer(c_int), intent(out) :: cnt(F%n_active, F%n_replicas)
        cnt = F%cnt
    end subroutine mfarm_get_counts

    ! host mirror of one molecule: active-type index ia (0-based), slot (0-based), replica (0-based)
    subroutine mfarm_get_molecule(replica, ia, slot, com, off) bind(C, name="mfarm_get_molecule")
        integer(c_int), value :: replica, ia, slot
        real(c_double), intent(out) :: com(3), off(3, F%max_n1)
        real(c_double), allocatable :: c_all(:, :), o_all(:, :, :)
        integer(c_int) :: nm, rc
        if (F%device_build) then
            allocate(c_all(3, F%cap(ia + 1)), o_all(3, F%n1(ia + 1), F%cap(ia + 1)))
            rc = mgpu_replica_get_frames(F%engine, replica, int(F%res_type(ia + 1), c_int), nm, c_all, o_all)
            com = c_all(:, slot + 1)
            off = 0.0_real64
            off(:, 1:F%n1(ia + 1)) = o_all(:, :, slot + 1)
            return
        end if
        com = F%mol(1:3, F%first(ia + 1) + slot + 1, replica + 1)
        off = reshape(F%mol(4:, F%first(ia + 1) + slot + 1, replica + 1), [3, F%max_n1])
    end subroutine mfarm_get_molecule

end module mc_farm
