!=======================================================================
! TEST INFRASTRUCTURE (oracle/ref) -- tiny record-stream writer used by the
! reference harness; read back by oracle/ref/run_ref.py:read_dump.
!=======================================================================
module evp_dumpio

  use ice_kinds_mod
  implicit none
  public

  integer, parameter :: dump_unit = 77
  logical            :: dump_open = .false.

contains

  subroutine dump_begin(fname)
    character(len=*), intent(in) :: fname
    open(unit=dump_unit, file=fname, form='unformatted', access='stream', status='replace')
    dump_open = .true.
  end subroutine dump_begin

  subroutine dump_end
    if (dump_open) close(dump_unit)
    dump_open = .false.
  end subroutine dump_end

  ! record = name(32 chars) | type code (1=r8, 2=i4) | d1 d2 d3 (int32) | payload
  subroutine dump_hdr(name, tcode, d1, d2, d3)
    character(len=*), intent(in) :: name
    integer(int_kind), intent(in) :: tcode, d1, d2, d3
    character(len=32) :: nm
    nm = name
    write(dump_unit) nm, tcode, d1, d2, d3
  end subroutine dump_hdr

  subroutine dump_r8_3d(name, a, nb)
    character(len=*), intent(in) :: name
    real(dbl_kind), dimension(:,:,:), intent(in) :: a
    integer(int_kind), intent(in) :: nb          ! number of real blocks (<= size(a,3))
    call dump_hdr(name, 1, size(a,1), size(a,2), nb)
    write(dump_unit) a(:,:,1:nb)
  end subroutine dump_r8_3d

  subroutine dump_l_3d(name, a, nb)
    character(len=*), intent(in) :: name
    logical(log_kind), dimension(:,:,:), intent(in) :: a
    integer(int_kind), intent(in) :: nb
    integer(int_kind), allocatable :: ia(:,:,:)
    allocate(ia(size(a,1),size(a,2),nb))
    ia = 0
    where (a(:,:,1:nb)) ia = 1
    call dump_hdr(name, 2, size(a,1), size(a,2), nb)
    write(dump_unit) ia
    deallocate(ia)
  end subroutine dump_l_3d

  subroutine dump_i4_1d(name, a)
    character(len=*), intent(in) :: name
    integer(int_kind), dimension(:), intent(in) :: a
    call dump_hdr(name, 2, size(a), 1, 1)
    write(dump_unit) a
  end subroutine dump_i4_1d

  subroutine dump_r8_1d(name, a)
    character(len=*), intent(in) :: name
    real(dbl_kind), dimension(:), intent(in) :: a
    call dump_hdr(name, 1, size(a), 1, 1)
    write(dump_unit) a
  end subroutine dump_r8_1d

  ! a module-private array of ice_dyn_evp, located by evp_peek.c
  subroutine dump_peek(name, which, n1, n2, n3, nb)
    use, intrinsic :: iso_c_binding, only: c_ptr, c_f_pointer, c_int, c_associated
    character(len=*), intent(in) :: name
    integer(int_kind), intent(in) :: which, n1, n2, n3, nb
    interface
       function evp_peek_base(w) bind(C, name='evp_peek_base') result(p)
         use, intrinsic :: iso_c_binding
         integer(c_int), value :: w
         type(c_ptr) :: p
       end function evp_peek_base
    end interface
    type(c_ptr) :: p
    real(dbl_kind), pointer :: pk(:,:,:)
    p = evp_peek_base(int(which, c_int))
    if (.not. c_associated(p)) stop 'evp_peek_base: unknown or unallocated array'
    call c_f_pointer(p, pk, [n1, n2, n3])
    call dump_r8_3d(name, pk, nb)
  end subroutine dump_peek

  ! the same array as a Fortran pointer (drop-in check of the C-grid entry: the harness hands the module-private
  ! operands of the loop to dyn_evp_hip_cgrid_run exactly as a call from inside evp() would)
  function peek_array(which, n1, n2, n3) result(pk)
    use, intrinsic :: iso_c_binding, only: c_ptr, c_f_pointer, c_int, c_associated
    integer(int_kind), intent(in) :: which, n1, n2, n3
    real(dbl_kind), pointer :: pk(:,:,:)
    interface
       function evp_peek_base(w) bind(C, name='evp_peek_base') result(p)
         use, intrinsic :: iso_c_binding
         integer(c_int), value :: w
         type(c_ptr) :: p
       end function evp_peek_base
    end interface
    type(c_ptr) :: p
    p = evp_peek_base(int(which, c_int))
    if (.not. c_associated(p)) stop 'evp_peek_base: unknown or unallocated array'
    call c_f_pointer(p, pk, [n1, n2, n3])
  end function peek_array

end module evp_dumpio
