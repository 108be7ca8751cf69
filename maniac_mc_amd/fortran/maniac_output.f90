!===============================================================================
! maniac_output -- state of ONE Markov chain in the reference's terms, and the reference's output
! surface for it (SURVEY 8(f) row 3): log.maniac status table, energy.dat, number_<res>.dat,
! moves.dat, trajectory.lammpstrj / reservoir.lammpstrj, restartable topology.data.
!
! Every record is produced with the edit descriptors (or the list-directed form) the reference uses,
! by the same Fortran runtime, so the files compare byte for byte with the reference's:
!   WriteLAMMPSTRJ        src/write_utils.f90:12-93
!   WriteEnergyAndCount   src/write_utils.f90:95-190   (incl. the Rotate_Moves column, which prints
!                                                       counter%deletions, :186)
!   WriteLAMMPSData       src/write_utils.f90:192-414
!   UpdateFiles           src/write_utils.f90:420-434
!   LogMessage / BoxLine / LogStartMC / PrintStatus / FinalReport / PrintTerminationMessage
!                         src/output_utils.f90:30-275
! The state lives in explicit derived types instead of the reference's module globals
! (src/simulation_state.f90:85-190); per residue type it holds what primary%mol_com /
! primary%site_offset / primary%num_residues hold there.
!===============================================================================
module maniac_output

    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: real64

    implicit none

    private
    public :: residue_block, box_block, chain_block
    public :: log_line, log_box_line, log_rule, log_start_mc, log_status, log_final_report
    public :: write_trajectory, write_energy_and_count, write_topology, update_files, wrap_into_box
    public :: KB_KCALMOL_OUT, BOX_WIDTH, mout_format_fixed

    integer, parameter :: BOX_WIDTH = 78                                  ! src/parameters.f90:25
    real(real64), parameter :: KB_KCALMOL_OUT = 0.0019872041_real64       ! src/constants.f90 KB_kcalmol
    character(len=*), parameter :: TOPOLOGY_NAME = 'topology.data'        ! src/parameters.f90:32

    interface
        ! include/maniac_gpu.h: the atom records of the two big files, formatted outside the Fortran runtime (same bytes)
        function mgpu_append_atom_records(path, n, first_serial, mol, atype, charge, xyz) &
                bind(C, name="mgpu_append_atom_records") result(rc)
            import :: c_char, c_int, c_ptr, c_double
            character(kind=c_char), intent(in) :: path(*)
            integer(c_int), value :: n, first_serial
            type(c_ptr), value :: mol, charge               ! c_null_ptr: trajectory records
            integer(c_int), intent(in) :: atype(*)
            real(c_double), intent(in) :: xyz(3, *)
            integer(c_int) :: rc
        end function
    end interface

    ! energy components, in the order of type(energy_state) as the writers print them
    integer, parameter, public :: IE_NONC = 1, IE_COUL = 2, IE_RECIP = 3, IE_SELF = 4, IE_INTRA = 5, IE_TOTAL = 6
    ! counters: trial / accepted per move type (src/simulation_state.f90:19-31)
    integer, parameter, public :: C_TRIAL_T = 1, C_T = 2, C_TRIAL_R = 3, C_R = 4, C_TRIAL_C = 5, C_C = 6, &
                                  C_TRIAL_D = 7, C_D = 8

    type :: residue_block
        character(len=10) :: name = ''
        integer :: n1 = 0, active = 0, cap = 0, count = 0
        integer, allocatable :: atom_type(:)            ! (n1)        primary%atom_types(t, :)
        real(real64), allocatable :: charge(:)          ! (n1)        primary%atom_charges(t, :)
        real(real64), allocatable :: com(:, :)          ! (3, cap)    primary%mol_com(:, t, :)
        real(real64), allocatable :: off(:, :, :)       ! (3, n1, cap) primary%site_offset(:, t, :, :)
        real(real64) :: fugacity = 0.0_real64           ! molecules per cubic Angstrom (after ConvertFugacity)
        integer :: n_bonded(4) = 0                      ! bonds, angles, dihedrals, impropers of one molecule
        integer, allocatable :: bonded(:, :, :)         ! (5, max(n_bonded), 4): type, local atom indices
    end type residue_block

    type :: box_block
        real(real64) :: matrix(3, 3) = 0, reciprocal(3, 3) = 0, lo(3) = 0, hi(3) = 0, tilt(3) = 0, volume = 0
        logical :: is_triclinic = .false.               ! a tilt line was present (box%is_triclinic)
        integer :: box_type = 1                         ! 1 cubic, 2 orthorhombic, 3 triclinic (box%type)
        integer :: num_atoms = 0
    end type box_block

    type :: chain_block
        type(c_ptr) :: engine = c_null_ptr
        integer :: n_res = 0, n_atom_types = 0
        type(residue_block), allocatable :: res(:)      ! primary
        type(residue_block), allocatable :: rsv(:)      ! reservoir (com / off / count only)
        type(box_block) :: box, rbox
        logical :: has_reservoir = .false.
        real(real64), allocatable :: masses(:)          ! box%site_masses_vector
        integer :: n_bonded_types(4) = 0                ! "<n> bond types" ... of the data file header
        logical :: any_bonded(4) = .false.              ! reservoir%num_bonds > 0, ... (header counts of the reservoir file)
        real(real64) :: temperature = 0, translation_step = 0, rotation_step = 0
        real(real64) :: p_translation = 0, p_rotation = 0
        logical :: recalibrate = .false.
        real(real64) :: energy(6) = 0
        integer :: counter(8) = 0
        integer :: current_block = 0, nb_block = 0, nb_step = 0
        character(len=512) :: outdir = ''
        integer :: log_unit = 10                        ! out_unit, src/simulation_state.f90:16
    end type chain_block

contains

    ! Test hook: x(i) through the Fortran runtime's Fw.d editing, w characters each, into out -- what
    ! mgpu_append_atom_records' own conversion must reproduce byte for byte (tests/test_host_setup.py)
    subroutine mout_format_fixed(n, x, w, d, out) bind(C, name="mout_format_fixed")
        integer(c_int), value :: n, w, d
        real(c_double), intent(in) :: x(n)
        character(kind=c_char), intent(out) :: out(w, n)
        character(len=32) :: fmt
        character(len=64) :: field
        integer :: i, k
        write(fmt, '(A,I0,A,I0,A)') '(F', w, '.', d, ')'
        do i = 1, n
            write(field, fmt) x(i)
            do k = 1, w
                out(k, i) = field(k:k)
            end do
        end do
    end subroutine mout_format_fixed

    !---------------------------------------------------------------------------
    ! log.maniac
    !---------------------------------------------------------------------------
    subroutine log_line(ch, msg)
        type(chain_block), intent(in) :: ch
        character(*), intent(in) :: msg
        write(ch%log_unit, *) trim(msg)                 ! list-directed, like LogMessage (output_utils.f90:30-36)
        flush(ch%log_unit)
    end subroutine log_line

    subroutine log_rule(ch)
        type(chain_block), intent(in) :: ch
        call log_line(ch, '+' // repeat('-', BOX_WIDTH - 2) // '+')
    end subroutine log_rule

    subroutine log_box_line(ch, text)
        type(chain_block), intent(in) :: ch
        character(*), intent(in) :: text
        character(len=BOX_WIDTH - 4) :: body
        body = adjustl(text)                            ! blank padded / truncated to the box width
        call log_line(ch, '| ' // body // ' |')
    end subroutine log_box_line

    subroutine log_start_mc(ch)
        type(chain_block), intent(in) :: ch
        call log_line(ch, '')
        call log_rule(ch)
        call log_box_line(ch, 'Started Monte Carlo Loop')
        call log_rule(ch)
        call log_line(ch, '')
    end subroutine log_start_mc

    ! PrintStatus (output_utils.f90:154-219)
    subroutine log_status(ch)
        type(chain_block), intent(in) :: ch
        character(len=1024) :: msg
        character(len=64) :: item
        real(real64) :: e_k(5), e_tot, e_coul, e_long
        integer :: t

        call log_line(ch, '')
        msg = '  Energy report | Active molecules: '
        do t = 1, ch%n_res
            if (ch%res(t)%count /= 0 .and. ch%res(t)%active == 1) then
                write(item, '(A,"=",I0)') trim(ch%res(t)%name), ch%res(t)%count
                msg = trim(msg) // ' ' // trim(item)
            end if
        end do
        call log_line(ch, msg)

        e_k(IE_RECIP) = ch%energy(IE_RECIP) * KB_KCALMOL_OUT
        e_k(IE_NONC) = ch%energy(IE_NONC) * KB_KCALMOL_OUT
        e_k(IE_COUL) = ch%energy(IE_COUL) * KB_KCALMOL_OUT
        e_k(IE_SELF) = ch%energy(IE_SELF) * KB_KCALMOL_OUT
        e_k(IE_INTRA) = ch%energy(IE_INTRA) * KB_KCALMOL_OUT
        e_tot = e_k(IE_NONC) + e_k(IE_RECIP) + e_k(IE_COUL) + e_k(IE_SELF) + e_k(IE_INTRA)
        e_coul = e_k(IE_COUL) + e_k(IE_INTRA)
        e_long = e_k(IE_RECIP) + e_k(IE_SELF)

        write(msg, '(A10,1X,A14,1X,A14,1X,A14,1X,A14,2X,A10,2X,A10,2X,A20)') &
            'Step', 'TotEng', 'E_vdwl', 'E_coul', 'E_long', 'TransStep', 'RotAngle', 'MC (acc/trial)'
        call log_line(ch, msg)
        write(msg, '(I10,1X,F14.4,1X,F14.4,1X,F14.4,1X,F14.4,2X,F10.4,2X,F10.4,2X,' // &
                   ' "T(",I0,"/",I0,") R(",I0,"/",I0,") C(",I0,"/",I0,") D(",I0,"/",I0,")")') &
            ch%current_block, e_tot, e_k(IE_NONC), e_coul, e_long, ch%translation_step, ch%rotation_step, &
            ch%counter(C_T), ch%counter(C_TRIAL_T), ch%counter(C_R), ch%counter(C_TRIAL_R), &
            ch%counter(C_C), ch%counter(C_TRIAL_C), ch%counter(C_D), ch%counter(C_TRIAL_D)
        call log_line(ch, msg)
    end subroutine log_status

    ! FinalReport + CloseOutput / PrintTerminationMessage (output_utils.f90:221-275, :19-24, :62-107)
    subroutine log_final_report(ch)
        type(chain_block), intent(in) :: ch
        character(len=1024) :: msg
        character(len=256) :: line
        real(real64) :: e_recip, e_nonc, e_coulomb, e_self, e_intra, e_tot, e_coul, e_long

        e_recip = ch%energy(IE_RECIP) * KB_KCALMOL_OUT
        e_nonc = ch%energy(IE_NONC) * KB_KCALMOL_OUT
        e_coulomb = ch%energy(IE_COUL) * KB_KCALMOL_OUT
        e_self = ch%energy(IE_SELF) * KB_KCALMOL_OUT
        e_intra = ch%energy(IE_INTRA) * KB_KCALMOL_OUT
        e_tot = e_nonc + e_recip + e_coulomb + e_self + e_intra
        e_coul = e_coulomb + e_intra
        e_long = e_recip + e_self

        call log_line(ch, '')
        call log_rule(ch)
        call log_box_line(ch, 'Final Energy Report')
        call log_box_line(ch, '')
        call log_box_line(ch, '  Step        TotEng        E_vdwl        E_coul        E_long')
        write(msg, '(I10,1X,F15.6,1X,F15.6,1X,F15.6,1X,F15.6)') ch%current_block, e_tot, e_nonc, e_coul, e_long
        call log_box_line(ch, trim(msg))
        call log_box_line(ch, '')
        call log_rule(ch)
        call log_line(ch, '')

        call log_line(ch, '')
        call log_rule(ch)
        call log_box_line(ch, 'MANIAC-MC Simulation Completed')
        call log_box_line(ch, '')
        write(line, '(A,I8,A,I8)') '  Translations (Trial/Accepted): ', ch%counter(C_TRIAL_T), ' / ', ch%counter(C_T)
        call log_box_line(ch, trim(line))
        write(line, '(A,I8,A,I8)') '  Rotations    (Trial/Accepted): ', ch%counter(C_TRIAL_R), ' / ', ch%counter(C_R)
        call log_box_line(ch, trim(line))
        write(line, '(A,I8,A,I8)') '  Creations    (Trial/Accepted): ', ch%counter(C_TRIAL_C), ' / ', ch%counter(C_C)
        call log_box_line(ch, trim(line))
        write(line, '(A,I8,A,I8)') '  Deletions    (Trial/Accepted): ', ch%counter(C_TRIAL_D), ' / ', ch%counter(C_D)
        call log_box_line(ch, trim(line))
        call log_box_line(ch, '')
        call log_box_line(ch, 'All output files have been written to:')
        call log_box_line(ch, trim(ch%outdir))
        call log_rule(ch)
        call log_line(ch, '')
    end subroutine log_final_report

    !---------------------------------------------------------------------------
    ! WrapIntoBox (geometry_utils.f90:225-263): [-L/2, L/2] per axis, or [-0.5, 0.5) in fractional space
    !---------------------------------------------------------------------------
    subroutine wrap_into_box(pos, box)
        real(real64), intent(inout) :: pos(3)
        type(box_block), intent(in) :: box
        real(real64) :: frac(3)
        integer :: d
        if (box%box_type == 1 .or. box%box_type == 2) then
            do d = 1, 3
                pos(d) = pos(d) - box%matrix(d, d) * nint(pos(d) / box%matrix(d, d))
            end do
        else if (box%box_type == 3) then
            frac = matmul(box%reciprocal, pos)
            do d = 1, 3
                frac(d) = frac(d) - nint(frac(d))
            end do
            pos = matmul(box%matrix, frac)
        end if
    end subroutine wrap_into_box

    !---------------------------------------------------------------------------
    ! trajectory.lammpstrj / reservoir.lammpstrj
    !---------------------------------------------------------------------------
    subroutine write_trajectory(ch, res, box, filename, append)
        type(chain_block), intent(in) :: ch
        type(residue_block), intent(in) :: res(:)
        type(box_block), intent(in) :: box
        character(*), intent(in) :: filename
        logical, intent(in) :: append
        integer, parameter :: u = 18
        integer :: t, m, a, serial, i
        real(real64) :: com(3), pos(3)
        integer(c_int), allocatable :: ty(:)
        real(real64), allocatable :: xyz(:, :)

        if (append) then
            open(unit=u, file=trim(ch%outdir) // filename, status='unknown', action='write', position='append')
        else
            open(unit=u, file=trim(ch%outdir) // filename, status='unknown', action='write', position='asis')
        end if
        write(u, '(A)') 'ITEM: TIMESTEP'
        write(u, '(I10)') ch%nb_block                   ! the reference prints input%nb_block in every frame
        write(u, '(A)') 'ITEM: NUMBER OF ATOMS'
        write(u, '(I10)') box%num_atoms
        write(u, '(A)') 'ITEM: BOX BOUNDS pp pp pp'
        do a = 1, 3
            write(u, '(F15.8,1X,F15.8)') -box%matrix(a, a) / 2, box%matrix(a, a) / 2
        end do
        write(u, '(A)') 'ITEM: ATOMS id type x y z'
        serial = 0
        do t = 1, ch%n_res
            serial = serial + res(t)%count * ch%res(t)%n1
        end do
        allocate(ty(serial), xyz(3, serial))
        serial = 0
        do t = 1, ch%n_res
            do m = 1, res(t)%count
                com = res(t)%com(:, m)
                if (ch%res(t)%active == 1) call wrap_into_box(com, box)     ! active molecules: wrap the COM
                do a = 1, ch%res(t)%n1
                    serial = serial + 1
                    pos = com + res(t)%off(:, a, m)
                    if (ch%res(t)%active == 0) call wrap_into_box(pos, box)  ! inactive structure: wrap every atom
                    ty(serial) = ch%res(t)%atom_type(a)
                    xyz(:, serial) = pos
                end do
            end do
        end do
        ! the atom records (write_utils.f90:86's edit descriptors) are appended by the engine library's exact formatter: the
        ! runtime's Fw.d conversion costs 8 ms per frame at 10 125 atoms; should that fail, by one formatted write here
        close(u)
        if (serial > 0) then
            if (mgpu_append_atom_records(trim(ch%outdir) // filename // c_null_char, int(serial, c_int), 1_c_int, c_null_ptr, ty, &
                                         c_null_ptr, xyz) /= 0) then
                open(unit=u, file=trim(ch%outdir) // filename, status='old', action='write', position='append')
                write(u, '((I6,1X,I4,3(1X,F12.7)))') (i, ty(i), xyz(1, i), xyz(2, i), xyz(3, i), i = 1, serial)
                close(u)
            end if
        end if
    end subroutine write_trajectory

    !---------------------------------------------------------------------------
    ! energy.dat, number_<res>.dat, moves.dat -- one record per block; block 0 creates the files
    !---------------------------------------------------------------------------
    subroutine write_energy_and_count(ch)
        type(chain_block), intent(in) :: ch
        integer, parameter :: u_e = 18, u_n = 19, u_m = 20
        character(len=8) :: status
        real(real64) :: k(6)
        integer :: t

        if (ch%current_block == 0) then
            status = 'REPLACE'
        else
            status = 'OLD'
        end if
        k = ch%energy * KB_KCALMOL_OUT

        open(unit=u_e, file=trim(ch%outdir) // 'energy.dat', status=status, action='write', position='append')
        if (ch%current_block == 0) write(u_e, '(A)') '#    block        total        recipCoulomb' // &
            '     non-coulomb      coulomb     ewald_self    intramolecular-coulomb'
        write(u_e, '(I10, 1X, F16.6, 1X, F16.6, 1X, F16.6, 1X, F16.6, 1X, F16.6, 1X, F16.6)') &
            ch%current_block, k(IE_TOTAL), k(IE_RECIP), k(IE_NONC), k(IE_COUL), k(IE_SELF), k(IE_INTRA)
        close(u_e)

        do t = 1, ch%n_res
            if (ch%res(t)%count /= 0 .and. ch%res(t)%active == 1) then
                open(unit=u_n, file=trim(ch%outdir) // 'number_' // trim(ch%res(t)%name) // '.dat', status=status, &
                     action='write', position='append')
                if (ch%current_block == 0) write(u_n, '(A)') '# Block   Active_Molecules'
                write(u_n, '(I10, 1X, I10)') ch%current_block, ch%res(t)%count
                close(u_n)
            end if
        end do

        open(unit=u_m, file=trim(ch%outdir) // 'moves.dat', status=status, action='write', position='append')
        if (ch%current_block == 0) write(u_m, '(A)') &
            '# Block   Trial_Trans   Trans_Moves   Trial_Create   Create_Moves   ' // &
            'Trial_Delete   Delete_Moves   Trial_Rotate   Rotate_Moves   ' // &
            'Trial_BigMove   Big_Moves'
        ! the last column repeats the accepted deletions, as the reference writes it (write_utils.f90:186)
        write(u_m, '(I12, 1X, I12, 1X, I12, 1X, I12, 1X, I12, 1X, I12, 1X,' // 'I12, 1X, I12, 1X, I12)') &
            ch%current_block, ch%counter(C_TRIAL_T), ch%counter(C_T), ch%counter(C_TRIAL_C), ch%counter(C_C), &
            ch%counter(C_TRIAL_D), ch%counter(C_D), ch%counter(C_TRIAL_R), ch%counter(C_D)
        close(u_m)
    end subroutine write_energy_and_count

    !---------------------------------------------------------------------------
    ! topology.data: LAMMPS data file (atom_style full) of the primary box, rewritten every block
    !---------------------------------------------------------------------------
    subroutine write_topology(ch)
        type(chain_block), intent(in) :: ch
        integer, parameter :: u = 19
        character(len=9), parameter :: section(4) = [character(len=9) :: 'Bonds', 'Angles', 'Dihedrals', 'Impropers']
        integer :: t, m, a, k, kind, serial, mol, total(4), first_atom, n_members, c, i
        real(real64) :: pos(3)
        integer(c_int), allocatable, target :: mols(:)
        integer(c_int), allocatable :: tys(:)
        real(real64), allocatable, target :: qs(:)
        real(real64), allocatable :: xyz(:, :)

        total = 0
        do t = 1, ch%n_res
            total = total + ch%res(t)%count * ch%res(t)%n_bonded
        end do

        open(unit=u, file=trim(ch%outdir) // TOPOLOGY_NAME, status='replace', action='write')
        write(u, *) '! LAMMPS data file (atom_style full)'
        write(u, *) ch%box%num_atoms, ' atoms'
        write(u, *) ch%n_atom_types, ' atom types'
        write(u, *) total(1), ' bonds'
        write(u, *) ch%n_bonded_types(1), ' bond types'
        write(u, *) total(2), ' angles'
        write(u, *) ch%n_bonded_types(2), ' angle types'
        write(u, *) total(3), ' dihedrals'
        write(u, *) ch%n_bonded_types(3), ' dihedral types'
        write(u, *) total(4), ' impropers'
        write(u, *) ch%n_bonded_types(4), ' improper types'
        write(u, *)
        write(u, '(2(F15.8,1X))', advance='no') ch%box%lo(1), ch%box%hi(1)
        write(u, '(A)') 'xlo xhi'
        write(u, '(2(F15.8,1X))', advance='no') ch%box%lo(2), ch%box%hi(2)
        write(u, '(A)') 'ylo yhi'
        write(u, '(2(F15.8,1X))', advance='no') ch%box%lo(3), ch%box%hi(3)
        write(u, '(A)') 'zlo zhi'
        if (ch%box%is_triclinic) then
            write(u, '(3(F15.8,1X))') ch%box%tilt(1), ch%box%tilt(2), ch%box%tilt(3)
            write(u, '(A)') 'xy xz yz'
        end if
        write(u, *)
        write(u, *) 'Masses'
        write(u, *)
        do k = 1, ch%n_atom_types
            write(u, '(I5, 1X, F12.6)') k, ch%masses(k)
        end do
        write(u, *)
        write(u, *) 'Atoms'
        write(u, *)
        serial = 0
        do t = 1, ch%n_res
            serial = serial + ch%res(t)%count * ch%res(t)%n1
        end do
        allocate(mols(serial), tys(serial), qs(serial), xyz(3, serial))
        serial = 0
        mol = 0
        do t = 1, ch%n_res
            do m = 1, ch%res(t)%count
                mol = mol + 1
                do a = 1, ch%res(t)%n1
                    serial = serial + 1
                    pos = ch%res(t)%com(:, m) + ch%res(t)%off(:, a, m)
                    ! active molecules stay whole across the boundary; only the inactive structure is wrapped
                    if (ch%res(t)%active == 0) call wrap_into_box(pos, ch%box)
                    mols(serial) = mol
                    tys(serial) = ch%res(t)%atom_type(a)
                    qs(serial) = ch%res(t)%charge(a)
                    xyz(:, serial) = pos
                end do
            end do
        end do
        ! the section's records (write_utils.f90:297-300's edit descriptors) by the engine library's exact formatter, as in
        ! write_trajectory; the unit is closed around the append and reopened for the sections that follow
        if (serial > 0) then
            close(u)
            if (mgpu_append_atom_records(trim(ch%outdir) // TOPOLOGY_NAME // c_null_char, int(serial, c_int), 1_c_int, c_loc(mols), tys, &
                                         c_loc(qs), xyz) /= 0) then
                open(unit=u, file=trim(ch%outdir) // TOPOLOGY_NAME, status='old', action='write', position='append')
                write(u, '((I6,1X,I6,1X,I4,1X,F12.8,3(1X,F12.7)))') (i, mols(i), tys(i), qs(i), xyz(1, i), xyz(2, i), xyz(3, i), i = 1, serial)
            else
                open(unit=u, file=trim(ch%outdir) // TOPOLOGY_NAME, status='old', action='write', position='append')
            end if
        end if

        do kind = 1, 4
            ! primary%num_bonds > 0 .or. reservoir%num_bonds > 0 (write_utils.f90:336), the primary count being
            ! the one just recomputed above
            if (.not. (total(kind) > 0 .or. ch%any_bonded(kind))) cycle
            n_members = merge(2, merge(3, 4, kind == 2), kind == 1)
            write(u, *)
            write(u, *) trim(section(kind))
            write(u, *)
            serial = 1
            first_atom = 0
            do t = 1, ch%n_res
                do m = 1, ch%res(t)%count
                    do k = 1, ch%res(t)%n_bonded(kind)
                        select case (n_members)
                        case (2)
                            write(u, *) serial, ch%res(t)%bonded(1, k, kind), &
                                (first_atom + ch%res(t)%bonded(1 + c, k, kind), c = 1, 2)
                        case (3)
                            write(u, *) serial, ch%res(t)%bonded(1, k, kind), &
                                (first_atom + ch%res(t)%bonded(1 + c, k, kind), c = 1, 3)
                        case default
                            write(u, *) serial, ch%res(t)%bonded(1, k, kind), &
                                (first_atom + ch%res(t)%bonded(1 + c, k, kind), c = 1, 4)
                        end select
                        serial = serial + 1
                    end do
                    first_atom = first_atom + ch%res(t)%n1
                end do
            end do
        end do
        close(u)
    end subroutine write_topology

    ! UpdateFiles (write_utils.f90:420-434)
    subroutine update_files(ch, later_step)
        type(chain_block), intent(in) :: ch
        logical, intent(in) :: later_step
        call write_trajectory(ch, ch%res, ch%box, 'trajectory.lammpstrj', later_step)
        if (ch%has_reservoir) call write_trajectory(ch, ch%rsv, ch%rbox, 'reservoir.lammpstrj', later_step)
        call write_energy_and_count(ch)
        call write_topology(ch)
    end subroutine update_files

end module maniac_output
