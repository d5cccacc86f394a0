! basic_usage.f90 -- the smallest complete Fortran program on the library: create a grid descriptor, query the
! pencils, allocate device buffers, run one X -> Y -> Z -> Y -> X transpose cycle and one halo update.
! Counterpart of examples/c/basic_usage.c (and of the reference's examples/fortran/basic_usage); device memory is
! addressed through ordinary Fortran pointers whose target lives on the GPU (see INTEGRATION.md, Fortran).
!
! build:  make -C fortran && amdflang -Ifortran/build examples/fortran/basic_usage.f90 -o basic_usage_f \
!           -Lfortran/build -lcudecomp_fort -Lcudecomp_amd/lib -lcudecomp -L/opt/rocm/lib -lamdhip64
! run:    ./basic_usage_f                         (one rank)
!         RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p ./basic_usage_f   (n ranks, or any launcher
!         that exports rank / size: torchrun, mpirun, srun)
program basic_usage
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64, real64
  use cudecomp
  implicit none

  interface
    function hipSetDevice(dev) bind(C, name="hipSetDevice") result(res)
      import
      integer(c_int), value :: dev
      integer(c_int) :: res
    end function hipSetDevice
    function hipGetDeviceCount(n) bind(C, name="hipGetDeviceCount") result(res)
      import
      integer(c_int) :: n
      integer(c_int) :: res
    end function hipGetDeviceCount
    function hipMemcpy(dst, src, bytes, kind) bind(C, name="hipMemcpy") result(res)
      import
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int), value :: kind
      integer(c_int) :: res
    end function hipMemcpy
    function hipDeviceSynchronize() bind(C, name="hipDeviceSynchronize") result(res)
      import
      integer(c_int) :: res
    end function hipDeviceSynchronize
  end interface

  integer, parameter :: MPI_COMM_WORLD_MPICH = int(z'44000000')  ! with a real MPI: `use mpi` and MPI_COMM_WORLD
  type(cudecompHandle) :: handle
  type(cudecompGridDesc) :: grid_desc
  type(cudecompGridDescConfig) :: config
  type(cudecompPencilInfo) :: px, py, pz
  real(real64), pointer, contiguous :: d_a(:), d_b(:), d_work(:)
  real(real64), allocatable, target :: h_in(:), h_out(:)
  integer(int64) :: n, work_t, work_h
  integer :: rank, nranks, ndev, ierr, halo(3)
  integer(int64) :: i
  character(len=32) :: env
  integer :: elen, estat

  rank = 0
  nranks = 1
  call get_environment_variable("RANK", env, elen, estat)
  if (estat == 0 .and. elen > 0) read (env(1:elen), *) rank
  call get_environment_variable("WORLD_SIZE", env, elen, estat)
  if (estat == 0 .and. elen > 0) read (env(1:elen), *) nranks

  ierr = hipGetDeviceCount(ndev)
  ierr = hipSetDevice(mod(rank, max(ndev, 1)))
  call ok(cudecompInit(handle, MPI_COMM_WORLD_MPICH), "cudecompInit")

  call ok(cudecompGridDescConfigSetDefaults(config), "config defaults")
  config%gdims = [64, 48, 40]
  config%pdims = [1, nranks]                              ! slabs; set [0,0] and pass autotune options to search
  config%transpose_comm_backend = CUDECOMP_TRANSPOSE_COMM_NCCL
  config%halo_comm_backend = CUDECOMP_HALO_COMM_NCCL
  config%transpose_axis_contiguous = [.true., .true., .true.]
  if (nranks > 1) then                                    ! ranks of this demo may share one GPU: use the xGMI
    config%transpose_comm_backend = CUDECOMP_TRANSPOSE_COMM_NVSHMEM   ! peer transport, RCCL needs a GPU per rank
    config%halo_comm_backend = CUDECOMP_HALO_COMM_NVSHMEM
  end if
  call ok(cudecompGridDescCreate(handle, grid_desc, config), "cudecompGridDescCreate")

  halo = [1, 1, 1]
  call ok(cudecompGetPencilInfo(handle, grid_desc, px, 1, halo), "pencil info x")
  call ok(cudecompGetPencilInfo(handle, grid_desc, py, 2), "pencil info y")
  call ok(cudecompGetPencilInfo(handle, grid_desc, pz, 3), "pencil info z")
  if (rank == 0) then
    print '(a,3i5,a,3i3)', " x-pencil shape (with halos):", px%shape, "  order:", px%order
    print '(a,3i5,a,3i3)', " y-pencil shape:             ", py%shape, "  order:", py%order
    print '(a,3i5,a,3i3)', " z-pencil shape:             ", pz%shape, "  order:", pz%order
  end if

  call ok(cudecompGetTransposeWorkspaceSize(handle, grid_desc, work_t), "transpose workspace")
  call ok(cudecompGetHaloWorkspaceSize(handle, grid_desc, 1, halo, work_h), "halo workspace")
  n = max(px%size, py%size, pz%size)
  call ok(cudecompMalloc(handle, grid_desc, d_a, n), "cudecompMalloc a")
  call ok(cudecompMalloc(handle, grid_desc, d_b, n), "cudecompMalloc b")
  call ok(cudecompMalloc(handle, grid_desc, d_work, max(work_t, work_h, 1_int64)), "cudecompMalloc work")

  allocate (h_in(px%size), h_out(px%size))
  do i = 1, px%size
    h_in(i) = real(rank, real64)*1.0d6 + real(i, real64)
  end do
  ierr = hipMemcpy(c_loc(d_a), c_loc(h_in), int(px%size*8, c_size_t), 1)

  ! the x-pencil carries halos, the others do not: say so per call
  call ok(cudecompTransposeXToY(handle, grid_desc, d_a, d_b, d_work, CUDECOMP_DOUBLE, input_halo_extents=halo), "XToY")
  call ok(cudecompTransposeYToZ(handle, grid_desc, d_b, d_b, d_work, CUDECOMP_DOUBLE), "YToZ (in place)")
  call ok(cudecompTransposeZToY(handle, grid_desc, d_b, d_b, d_work, CUDECOMP_DOUBLE), "ZToY (in place)")
  call ok(cudecompTransposeYToX(handle, grid_desc, d_b, d_a, d_work, CUDECOMP_DOUBLE, output_halo_extents=halo), "YToX")
  ! fill the halo cells of the x-pencil along y and z (dims 2 and 3), periodic in y only
  call ok(cudecompUpdateHalosX(handle, grid_desc, d_a, d_work, CUDECOMP_DOUBLE, halo, [.true., .true., .false.], 2), &
          "UpdateHalosX dim 2")
  call ok(cudecompUpdateHalosX(handle, grid_desc, d_a, d_work, CUDECOMP_DOUBLE, halo, [.true., .true., .false.], 3), &
          "UpdateHalosX dim 3")
  ierr = hipDeviceSynchronize()
  ierr = hipMemcpy(c_loc(h_out), c_loc(d_a), int(px%size*8, c_size_t), 2)

  ! the interior must be back unchanged (the halo cells now hold neighbours' values)
  if (interior_unchanged()) then
    print '(a,i0,a)', " rank ", rank, ": round trip OK"
  else
    print '(a,i0,a)', " rank ", rank, ": round trip MISMATCH"
    error stop 1
  end if

  call ok(cudecompFree(handle, grid_desc, d_a), "cudecompFree")
  call ok(cudecompFree(handle, grid_desc, d_b), "cudecompFree")
  call ok(cudecompFree(handle, grid_desc, d_work), "cudecompFree")
  call ok(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy")
  call ok(cudecompFinalize(handle), "cudecompFinalize")

contains

  subroutine ok(res, what)
    integer(c_int), intent(in) :: res
    character(len=*), intent(in) :: what
    if (res /= CUDECOMP_RESULT_SUCCESS) then
      print '(a,a,a,i0)', " ", what, " failed with ", res
      error stop 1
    end if
  end subroutine ok

  logical function interior_unchanged()
    integer :: i0, i1, i2
    integer(int64) :: idx
    interior_unchanged = .true.
    do i2 = 1 + halo(px%order(3)), px%shape(3) - halo(px%order(3))
      do i1 = 1 + halo(px%order(2)), px%shape(2) - halo(px%order(2))
        do i0 = 1 + halo(px%order(1)), px%shape(1) - halo(px%order(1))
          idx = i0 + int(px%shape(1), int64)*((i1 - 1) + int(px%shape(2), int64)*(i2 - 1))
          if (h_out(idx) /= h_in(idx)) interior_unchanged = .false.
        end do
      end do
    end do
  end function interior_unchanged

end program basic_usage
