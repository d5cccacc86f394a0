! cudecomp_m.f90 -- Fortran interface (module `cudecomp`) to the MI355X pencil-decomposition library.
!
! Mirrors the Fortran API of the reference (src/cudecomp_m.cuf:182-541 interfaces, :560-1086 wrappers) so a
! Fortran solver keeps its `use cudecomp` and its calls, but is plain Fortran 2008 + iso_c_binding compiled by
! amdflang -- no CUDA Fortran: there is no `device` attribute on AMD, so device buffers are ordinary Fortran
! arrays / pointers whose *address* is a device address (what hipfort's hipMalloc, `!$omp target data
! use_device_addr` or cudecompMalloc below hand out).  The library never dereferences them on the host.
!
! Conventions that differ from the C API, all inherited from the reference module:
!   * axes, dims and memory orders are ONE-based (x/y/z = 1/2/3): `axis`, `dim`, `halo_axis`,
!     `transpose_mem_order`, `pencil_info%order` (reference :605-648, :808-822, :1001);
!   * `pencil_info%lo/hi` are one-based global coordinates (reference :646-648);
!   * multi-dimensional members are column-major: transpose_mem_order(3,3) is (position, axis) and the autotune
!     halo/padding tables are (3,4) = (dimension, transpose op);
!   * optional halo / padding / stream arguments default to zero / the null stream.
! Streams are `integer(cudecomp_stream_kind)` holding a hipStream_t by value (hipfort users:
! `transfer(stream_c_ptr, 0_cudecomp_stream_kind)`).

module cudecomp_internal
  use, intrinsic :: iso_c_binding
  implicit none
  public

  ! C struct cudecompGridDescConfig_t (include/cudecomp.h), 104 bytes
  type, bind(c) :: cudecompGridDescConfig_v1
    integer(c_int64_t) :: struct_size
    integer(c_int32_t) :: magic
    integer(c_int32_t) :: version
    integer(c_int32_t) :: gdims(3)
    integer(c_int32_t) :: gdims_dist(3)
    integer(c_int32_t) :: pdims(2)
    integer(c_int32_t) :: rank_order
    integer(c_int32_t) :: transpose_comm_backend
    logical(c_bool) :: transpose_axis_contiguous(3)
    integer(c_int32_t) :: transpose_mem_order(3, 3)
    integer(c_int32_t) :: halo_comm_backend
  end type cudecompGridDescConfig_v1

  ! C struct cudecompGridDescAutotuneOptions_t, 320 bytes
  type, bind(c) :: cudecompGridDescAutotuneOptions_v1
    integer(c_int64_t) :: struct_size
    integer(c_int32_t) :: magic
    integer(c_int32_t) :: version
    integer(c_int32_t) :: n_warmup_trials
    integer(c_int32_t) :: n_trials
    integer(c_int32_t) :: grid_mode
    integer(c_int32_t) :: dtype
    logical(c_bool) :: allow_uneven_decompositions
    logical(c_bool) :: disable_mpi_backends
    logical(c_bool) :: disable_nccl_backends
    logical(c_bool) :: disable_nvshmem_backends
    real(c_double) :: skip_threshold
    logical(c_bool) :: autotune_transpose_backend
    logical(c_bool) :: transpose_use_inplace_buffers(4)
    real(c_double) :: transpose_op_weights(4)
    integer(c_int32_t) :: transpose_input_halo_extents(3, 4)
    integer(c_int32_t) :: transpose_output_halo_extents(3, 4)
    integer(c_int32_t) :: transpose_input_padding(3, 4)
    integer(c_int32_t) :: transpose_output_padding(3, 4)
    logical(c_bool) :: autotune_halo_backend
    integer(c_int32_t) :: halo_extents(3)
    logical(c_bool) :: halo_periods(3)
    integer(c_int32_t) :: halo_axis
    integer(c_int32_t) :: halo_padding(3)
  end type cudecompGridDescAutotuneOptions_v1

  ! C struct cudecompPencilInfo_t, 96 bytes
  type, bind(c) :: cudecompPencilInfo_v1
    integer(c_int64_t) :: struct_size
    integer(c_int32_t) :: magic
    integer(c_int32_t) :: version
    integer(c_int32_t) :: shape(3)
    integer(c_int32_t) :: lo(3)
    integer(c_int32_t) :: hi(3)
    integer(c_int32_t) :: order(3)
    integer(c_int32_t) :: halo_extents(3)
    integer(c_int32_t) :: padding(3)
    integer(c_int64_t) :: size
  end type cudecompPencilInfo_v1
end module cudecomp_internal

module cudecomp
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64, real32, real64
  use cudecomp_internal, only: cudecompGridDescConfig => cudecompGridDescConfig_v1, &
                               cudecompGridDescAutotuneOptions => cudecompGridDescAutotuneOptions_v1, &
                               cudecompPencilInfo => cudecompPencilInfo_v1
  implicit none
  private :: c_string_to_fortran

  integer, parameter :: cudecomp_stream_kind = c_intptr_t
  integer(c_int32_t), parameter, private :: STRUCT_VERSION = 1

  ! enum values as in include/cudecomp.h
  enum, bind(c)
    enumerator :: CUDECOMP_TRANSPOSE_COMM_MPI_P2P = 1
    enumerator :: CUDECOMP_TRANSPOSE_COMM_MPI_P2P_PL = 2
    enumerator :: CUDECOMP_TRANSPOSE_COMM_MPI_A2A = 3
    enumerator :: CUDECOMP_TRANSPOSE_COMM_NCCL = 4
    enumerator :: CUDECOMP_TRANSPOSE_COMM_NCCL_PL = 5
    enumerator :: CUDECOMP_TRANSPOSE_COMM_NVSHMEM = 6
    enumerator :: CUDECOMP_TRANSPOSE_COMM_NVSHMEM_PL = 7
    enumerator :: CUDECOMP_TRANSPOSE_COMM_NVSHMEM_SM = 8
  end enum
  enum, bind(c)
    enumerator :: CUDECOMP_HALO_COMM_MPI = 1
    enumerator :: CUDECOMP_HALO_COMM_MPI_BLOCKING = 2
    enumerator :: CUDECOMP_HALO_COMM_NCCL = 3
    enumerator :: CUDECOMP_HALO_COMM_NVSHMEM = 4
    enumerator :: CUDECOMP_HALO_COMM_NVSHMEM_BLOCKING = 5
  end enum
  enum, bind(c)
    enumerator :: CUDECOMP_AUTOTUNE_GRID_TRANSPOSE = 0
    enumerator :: CUDECOMP_AUTOTUNE_GRID_HALO = 1
  end enum
  enum, bind(c)
    enumerator :: CUDECOMP_RANK_ORDER_DEFAULT = 0
    enumerator :: CUDECOMP_RANK_ORDER_ROW_MAJOR = 1
    enumerator :: CUDECOMP_RANK_ORDER_COL_MAJOR = 2
  end enum
  enum, bind(c)
    enumerator :: CUDECOMP_FLOAT = -1
    enumerator :: CUDECOMP_DOUBLE = -2
    enumerator :: CUDECOMP_FLOAT_COMPLEX = -3
    enumerator :: CUDECOMP_DOUBLE_COMPLEX = -4
  end enum
  enum, bind(c)
    enumerator :: CUDECOMP_RESULT_SUCCESS = 0
    enumerator :: CUDECOMP_RESULT_INVALID_USAGE = 1
    enumerator :: CUDECOMP_RESULT_NOT_SUPPORTED = 2
    enumerator :: CUDECOMP_RESULT_INTERNAL_ERROR = 3
    enumerator :: CUDECOMP_RESULT_CUDA_ERROR = 4
    enumerator :: CUDECOMP_RESULT_CUTENSOR_ERROR = 5
    enumerator :: CUDECOMP_RESULT_MPI_ERROR = 6
    enumerator :: CUDECOMP_RESULT_NCCL_ERROR = 7
    enumerator :: CUDECOMP_RESULT_NVSHMEM_ERROR = 8
    enumerator :: CUDECOMP_RESULT_NVML_ERROR = 9
  end enum

  ! opaque handles: one C pointer each, passed by value
  type, bind(c) :: cudecompHandle
    type(c_ptr) :: member = c_null_ptr
  end type cudecompHandle
  type, bind(c) :: cudecompGridDesc
    type(c_ptr) :: member = c_null_ptr
  end type cudecompGridDesc

  ! ---- generic names -------------------------------------------------------------------------------------
  interface cudecompInit
    module procedure cudecompInit_comm_int, cudecompInit_comm_f08
  end interface cudecompInit
  interface cudecompMalloc
    module procedure cudecompMallocR4, cudecompMallocR8, cudecompMallocC4, cudecompMallocC8, cudecompMallocPtr
  end interface cudecompMalloc
  interface cudecompFree
    module procedure cudecompFreeR4, cudecompFreeR8, cudecompFreeC4, cudecompFreeC8, cudecompFreePtr
  end interface cudecompFree

  ! ---- the C entry points (include/cudecomp.h) ----------------------------------------------------------------
  interface
    function cudecompInit_FC(handle, mpi_comm) bind(C, name="cudecompInit_F") result(res)
      import
      type(cudecompHandle) :: handle
      integer(c_int), value :: mpi_comm
      integer(c_int) :: res
    end function cudecompInit_FC

    function cudecompFinalize(handle) bind(C, name="cudecompFinalize") result(res)
      import
      type(cudecompHandle), value :: handle
      integer(c_int) :: res
    end function cudecompFinalize

    function cudecompGridDescCreateC(handle, grid_desc, config, config_struct_size, config_version, options, &
                                     options_struct_size, options_version) &
      bind(C, name="cudecompGridDescCreateVersioned") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc) :: grid_desc
      type(c_ptr), value :: config
      integer(c_int64_t), value :: config_struct_size
      integer(c_int32_t), value :: config_version
      type(c_ptr), value :: options
      integer(c_int64_t), value :: options_struct_size
      integer(c_int32_t), value :: options_version
      integer(c_int) :: res
    end function cudecompGridDescCreateC

    function cudecompGridDescDestroy(handle, grid_desc) bind(C, name="cudecompGridDescDestroy") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      integer(c_int) :: res
    end function cudecompGridDescDestroy

    function cudecompGridDescConfigSetDefaultsC(config, struct_size, version) &
      bind(C, name="cudecompGridDescConfigSetDefaultsVersioned") result(res)
      import
      type(c_ptr), value :: config
      integer(c_int64_t), value :: struct_size
      integer(c_int32_t), value :: version
      integer(c_int) :: res
    end function cudecompGridDescConfigSetDefaultsC

    function cudecompGridDescAutotuneOptionsSetDefaultsC(options, struct_size, version) &
      bind(C, name="cudecompGridDescAutotuneOptionsSetDefaultsVersioned") result(res)
      import
      type(c_ptr), value :: options
      integer(c_int64_t), value :: struct_size
      integer(c_int32_t), value :: version
      integer(c_int) :: res
    end function cudecompGridDescAutotuneOptionsSetDefaultsC

    function cudecompGetPencilInfoC(handle, grid_desc, pencil_info, struct_size, version, axis, halo_extents, &
                                    padding) bind(C, name="cudecompGetPencilInfoVersioned") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: pencil_info
      integer(c_int64_t), value :: struct_size
      integer(c_int32_t), value :: version
      integer(c_int32_t), value :: axis
      integer(c_int32_t) :: halo_extents(3), padding(3)
      integer(c_int) :: res
    end function cudecompGetPencilInfoC

    function cudecompGetGridDescConfigC(handle, grid_desc, config, struct_size, version) &
      bind(C, name="cudecompGetGridDescConfigVersioned") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: config
      integer(c_int64_t), value :: struct_size
      integer(c_int32_t), value :: version
      integer(c_int) :: res
    end function cudecompGetGridDescConfigC

    function cudecompGetTransposeWorkspaceSize(handle, grid_desc, workspace_size) &
      bind(C, name="cudecompGetTransposeWorkspaceSize") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      integer(c_int64_t) :: workspace_size
      integer(c_int) :: res
    end function cudecompGetTransposeWorkspaceSize

    function cudecompGetHaloWorkspaceSizeC(handle, grid_desc, axis, halo_extents, workspace_size) &
      bind(C, name="cudecompGetHaloWorkspaceSize") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      integer(c_int32_t), value :: axis
      integer(c_int32_t) :: halo_extents(3)
      integer(c_int64_t) :: workspace_size
      integer(c_int) :: res
    end function cudecompGetHaloWorkspaceSizeC

    function cudecompGetDataTypeSize(dtype, dtype_size) bind(C, name="cudecompGetDataTypeSize") result(res)
      import
      integer(c_int), value :: dtype
      integer(c_int64_t) :: dtype_size
      integer(c_int) :: res
    end function cudecompGetDataTypeSize

    function cudecompMallocC(handle, grid_desc, buffer, buffer_size_bytes) bind(C, name="cudecompMalloc") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr) :: buffer
      integer(c_size_t), value :: buffer_size_bytes
      integer(c_int) :: res
    end function cudecompMallocC

    function cudecompFreeC(handle, grid_desc, buffer) bind(C, name="cudecompFree") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: buffer
      integer(c_int) :: res
    end function cudecompFreeC

    function cudecompTransposeCommBackendToStringC(comm_backend) &
      bind(C, name="cudecompTransposeCommBackendToString") result(res)
      import
      integer(c_int), value :: comm_backend
      type(c_ptr) :: res
    end function cudecompTransposeCommBackendToStringC

    function cudecompHaloCommBackendToStringC(comm_backend) &
      bind(C, name="cudecompHaloCommBackendToString") result(res)
      import
      integer(c_int), value :: comm_backend
      type(c_ptr) :: res
    end function cudecompHaloCommBackendToStringC

    function cudecompGetShiftedRankC(handle, grid_desc, axis, dim, displacement, periodic, shifted_rank) &
      bind(C, name="cudecompGetShiftedRank") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      integer(c_int32_t), value :: axis, dim, displacement
      logical(c_bool), value :: periodic
      integer(c_int32_t) :: shifted_rank
      integer(c_int) :: res
    end function cudecompGetShiftedRankC

    ! data-path entry points: buffers travel as raw addresses
    function cudecompTransposeXToY_C(handle, grid_desc, input, output, work, dtype, ihalo, ohalo, ipad, opad, &
                                     stream) bind(C, name="cudecompTransposeXToY") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: input, output, work
      integer(c_int), value :: dtype
      integer(c_int32_t) :: ihalo(3), ohalo(3), ipad(3), opad(3)
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function cudecompTransposeXToY_C

    function cudecompTransposeYToZ_C(handle, grid_desc, input, output, work, dtype, ihalo, ohalo, ipad, opad, &
                                     stream) bind(C, name="cudecompTransposeYToZ") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: input, output, work
      integer(c_int), value :: dtype
      integer(c_int32_t) :: ihalo(3), ohalo(3), ipad(3), opad(3)
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function cudecompTransposeYToZ_C

    function cudecompTransposeZToY_C(handle, grid_desc, input, output, work, dtype, ihalo, ohalo, ipad, opad, &
                                     stream) bind(C, name="cudecompTransposeZToY") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: input, output, work
      integer(c_int), value :: dtype
      integer(c_int32_t) :: ihalo(3), ohalo(3), ipad(3), opad(3)
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function cudecompTransposeZToY_C

    function cudecompTransposeYToX_C(handle, grid_desc, input, output, work, dtype, ihalo, ohalo, ipad, opad, &
                                     stream) bind(C, name="cudecompTransposeYToX") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: input, output, work
      integer(c_int), value :: dtype
      integer(c_int32_t) :: ihalo(3), ohalo(3), ipad(3), opad(3)
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function cudecompTransposeYToX_C

    function cudecompUpdateHalosX_C(handle, grid_desc, input, work, dtype, halo_extents, halo_periods, dim, &
                                    padding, stream) bind(C, name="cudecompUpdateHalosX") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: input, work
      integer(c_int), value :: dtype
      integer(c_int32_t) :: halo_extents(3)
      logical(c_bool) :: halo_periods(3)
      integer(c_int32_t), value :: dim
      integer(c_int32_t) :: padding(3)
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function cudecompUpdateHalosX_C

    function cudecompUpdateHalosY_C(handle, grid_desc, input, work, dtype, halo_extents, halo_periods, dim, &
                                    padding, stream) bind(C, name="cudecompUpdateHalosY") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: input, work
      integer(c_int), value :: dtype
      integer(c_int32_t) :: halo_extents(3)
      logical(c_bool) :: halo_periods(3)
      integer(c_int32_t), value :: dim
      integer(c_int32_t) :: padding(3)
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function cudecompUpdateHalosY_C

    function cudecompUpdateHalosZ_C(handle, grid_desc, input, work, dtype, halo_extents, halo_periods, dim, &
                                    padding, stream) bind(C, name="cudecompUpdateHalosZ") result(res)
      import
      type(cudecompHandle), value :: handle
      type(cudecompGridDesc), value :: grid_desc
      type(c_ptr), value :: input, work
      integer(c_int), value :: dtype
      integer(c_int32_t) :: halo_extents(3)
      logical(c_bool) :: halo_periods(3)
      integer(c_int32_t), value :: dim
      integer(c_int32_t) :: padding(3)
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function cudecompUpdateHalosZ_C

    function cudecomp_c_strlen(str) bind(C, name="strlen") result(n)
      import
      type(c_ptr), value :: str
      integer(c_size_t) :: n
    end function cudecomp_c_strlen
  end interface

contains

  ! ---- initialisation --------------------------------------------------------------------------------------
  ! `comm` is a Fortran MPI communicator handle (mpif.h / `use mpi`).  The non-MPI flavour of the library
  ! discovers its ranks from the launcher environment and only accepts the world communicator.
  function cudecompInit_comm_int(handle, comm) result(res)
    type(cudecompHandle) :: handle
    integer :: comm
    integer(c_int) :: res
    res = cudecompInit_FC(handle, int(comm, c_int))
  end function cudecompInit_comm_int

  ! `use mpi_f08` communicators: type(MPI_Comm) is a BIND(C) type with the single component MPI_VAL, so a
  ! structurally identical local definition names the same type.
  function cudecompInit_comm_f08(handle, comm) result(res)
    type, bind(c) :: MPI_Comm
      integer :: MPI_VAL
    end type MPI_Comm
    type(cudecompHandle) :: handle
    type(MPI_Comm) :: comm
    integer(c_int) :: res
    res = cudecompInit_FC(handle, int(comm%MPI_VAL, c_int))
  end function cudecompInit_comm_f08

  ! ---- configuration structs -------------------------------------------------------------------------------
  function cudecompGridDescConfigSetDefaults(config) result(res)
    type(cudecompGridDescConfig), target :: config
    integer(c_int) :: res
    res = cudecompGridDescConfigSetDefaultsC(c_loc(config), c_sizeof(config), STRUCT_VERSION)
    ! the C default is -1 ("unset") and stays -1 here: only valid orders are shifted to one-based on the way
    ! in and out, see cudecompGridDescCreate
  end function cudecompGridDescConfigSetDefaults

  function cudecompGridDescAutotuneOptionsSetDefaults(options) result(res)
    type(cudecompGridDescAutotuneOptions), target :: options
    integer(c_int) :: res
    res = cudecompGridDescAutotuneOptionsSetDefaultsC(c_loc(options), c_sizeof(options), STRUCT_VERSION)
    options%halo_axis = options%halo_axis + 1
  end function cudecompGridDescAutotuneOptionsSetDefaults

  ! config is in/out (autotuned pdims / backends are written back); options is optional
  function cudecompGridDescCreate(handle, grid_desc, config, options) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(cudecompGridDescConfig), target :: config
    type(cudecompGridDescAutotuneOptions), optional, target :: options
    integer(c_int) :: res

    ! one-based -> zero-based.  "unset" (-1 from SetDefaults) becomes -2, which the library treats like any
    ! negative entry: unset when the whole table is, invalid otherwise -- same as the reference module.
    config%transpose_mem_order = config%transpose_mem_order - 1
    if (present(options)) then
      options%halo_axis = options%halo_axis - 1
      res = cudecompGridDescCreateC(handle, grid_desc, c_loc(config), c_sizeof(config), STRUCT_VERSION, &
                                    c_loc(options), c_sizeof(options), STRUCT_VERSION)
      options%halo_axis = options%halo_axis + 1
    else
      res = cudecompGridDescCreateC(handle, grid_desc, c_loc(config), c_sizeof(config), STRUCT_VERSION, &
                                    c_null_ptr, 0_c_int64_t, 0_c_int32_t)
    end if
    config%transpose_mem_order = config%transpose_mem_order + 1
  end function cudecompGridDescCreate

  function cudecompGetGridDescConfig(handle, grid_desc, config) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(cudecompGridDescConfig), target :: config
    integer(c_int) :: res
    res = cudecompGetGridDescConfigC(handle, grid_desc, c_loc(config), c_sizeof(config), STRUCT_VERSION)
    config%transpose_mem_order = config%transpose_mem_order + 1
  end function cudecompGetGridDescConfig

  ! ---- geometry queries -------------------------------------------------------------------------------------
  function cudecompGetPencilInfo(handle, grid_desc, pencil_info, axis, halo_extents, padding) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(cudecompPencilInfo), target :: pencil_info  ! order, lo, hi come back one-based
    integer :: axis                                  ! 1/2/3 = x/y/z
    integer, optional :: halo_extents(3), padding(3)
    integer(c_int) :: res
    integer(c_int32_t) :: h(3), p(3)
    h = 0
    p = 0
    if (present(halo_extents)) h = int(halo_extents, c_int32_t)
    if (present(padding)) p = int(padding, c_int32_t)
    res = cudecompGetPencilInfoC(handle, grid_desc, c_loc(pencil_info), c_sizeof(pencil_info), STRUCT_VERSION, &
                                 int(axis - 1, c_int32_t), h, p)
    if (res /= CUDECOMP_RESULT_SUCCESS) return
    pencil_info%order = pencil_info%order + 1
    pencil_info%lo = pencil_info%lo + 1
    pencil_info%hi = pencil_info%hi + 1
  end function cudecompGetPencilInfo

  function cudecompGetHaloWorkspaceSize(handle, grid_desc, axis, halo_extents, workspace_size) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    integer :: axis
    integer :: halo_extents(3)
    integer(int64) :: workspace_size
    integer(c_int) :: res
    integer(c_int32_t) :: h(3)
    h = int(halo_extents, c_int32_t)
    res = cudecompGetHaloWorkspaceSizeC(handle, grid_desc, int(axis - 1, c_int32_t), h, workspace_size)
  end function cudecompGetHaloWorkspaceSize

  function cudecompGetShiftedRank(handle, grid_desc, axis, dim, displacement, periodic, shifted_rank) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    integer :: axis, dim, displacement
    logical :: periodic
    integer(c_int32_t) :: shifted_rank
    integer(c_int) :: res
    logical(c_bool) :: periodic_c
    periodic_c = periodic
    res = cudecompGetShiftedRankC(handle, grid_desc, int(axis - 1, c_int32_t), int(dim - 1, c_int32_t), &
                                  int(displacement, c_int32_t), periodic_c, shifted_rank)
  end function cudecompGetShiftedRank

  function cudecompTransposeCommBackendToString(comm_backend) result(res)
    integer :: comm_backend
    character(len=:), allocatable :: res
    call c_string_to_fortran(cudecompTransposeCommBackendToStringC(int(comm_backend, c_int)), res)
  end function cudecompTransposeCommBackendToString

  function cudecompHaloCommBackendToString(comm_backend) result(res)
    integer :: comm_backend
    character(len=:), allocatable :: res
    call c_string_to_fortran(cudecompHaloCommBackendToStringC(int(comm_backend, c_int)), res)
  end function cudecompHaloCommBackendToString

  subroutine c_string_to_fortran(cstr, fstr)
    type(c_ptr), intent(in) :: cstr
    character(len=:), allocatable, intent(out) :: fstr
    character(kind=c_char), pointer :: chars(:)
    integer :: i, n
    if (.not. c_associated(cstr)) then
      fstr = ""
      return
    end if
    n = int(cudecomp_c_strlen(cstr))
    call c_f_pointer(cstr, chars, [n])
    allocate (character(len=n) :: fstr)
    do i = 1, n
      fstr(i:i) = chars(i)
    end do
  end subroutine c_string_to_fortran

  ! ---- workspace allocation -----------------------------------------------------------------------------------
  ! buffer_size counts ELEMENTS of the pointer's type (reference :677-740).  The returned pointer addresses
  ! device memory: pass it to the transposes / halo updates / device kernels, never index it on the host.
  function cudecompMallocR4(handle, grid_desc, buffer, buffer_size) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    real(real32), pointer, contiguous :: buffer(:)
    integer(int64) :: buffer_size
    integer(c_int) :: res
    type(c_ptr) :: p
    p = c_null_ptr
    res = cudecompMallocC(handle, grid_desc, p, int(buffer_size * 4, c_size_t))
    call c_f_pointer(p, buffer, [buffer_size])
  end function cudecompMallocR4

  function cudecompMallocR8(handle, grid_desc, buffer, buffer_size) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    real(real64), pointer, contiguous :: buffer(:)
    integer(int64) :: buffer_size
    integer(c_int) :: res
    type(c_ptr) :: p
    p = c_null_ptr
    res = cudecompMallocC(handle, grid_desc, p, int(buffer_size * 8, c_size_t))
    call c_f_pointer(p, buffer, [buffer_size])
  end function cudecompMallocR8

  function cudecompMallocC4(handle, grid_desc, buffer, buffer_size) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    complex(real32), pointer, contiguous :: buffer(:)
    integer(int64) :: buffer_size
    integer(c_int) :: res
    type(c_ptr) :: p
    p = c_null_ptr
    res = cudecompMallocC(handle, grid_desc, p, int(buffer_size * 8, c_size_t))
    call c_f_pointer(p, buffer, [buffer_size])
  end function cudecompMallocC4

  function cudecompMallocC8(handle, grid_desc, buffer, buffer_size) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    complex(real64), pointer, contiguous :: buffer(:)
    integer(int64) :: buffer_size
    integer(c_int) :: res
    type(c_ptr) :: p
    p = c_null_ptr
    res = cudecompMallocC(handle, grid_desc, p, int(buffer_size * 16, c_size_t))
    call c_f_pointer(p, buffer, [buffer_size])
  end function cudecompMallocC8

  ! raw form for codes that keep device memory as type(c_ptr) (hipfort style); size in BYTES
  function cudecompMallocPtr(handle, grid_desc, buffer, buffer_size_bytes) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(c_ptr) :: buffer
    integer(int64) :: buffer_size_bytes
    integer(c_int) :: res
    buffer = c_null_ptr
    res = cudecompMallocC(handle, grid_desc, buffer, int(buffer_size_bytes, c_size_t))
  end function cudecompMallocPtr

  function cudecompFreeR4(handle, grid_desc, buffer) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    real(real32), pointer, contiguous :: buffer(:)
    integer(c_int) :: res
    res = cudecompFreeC(handle, grid_desc, c_loc(buffer))
    nullify (buffer)
  end function cudecompFreeR4

  function cudecompFreeR8(handle, grid_desc, buffer) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    real(real64), pointer, contiguous :: buffer(:)
    integer(c_int) :: res
    res = cudecompFreeC(handle, grid_desc, c_loc(buffer))
    nullify (buffer)
  end function cudecompFreeR8

  function cudecompFreeC4(handle, grid_desc, buffer) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    complex(real32), pointer, contiguous :: buffer(:)
    integer(c_int) :: res
    res = cudecompFreeC(handle, grid_desc, c_loc(buffer))
    nullify (buffer)
  end function cudecompFreeC4

  function cudecompFreeC8(handle, grid_desc, buffer) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    complex(real64), pointer, contiguous :: buffer(:)
    integer(c_int) :: res
    res = cudecompFreeC(handle, grid_desc, c_loc(buffer))
    nullify (buffer)
  end function cudecompFreeC8

  function cudecompFreePtr(handle, grid_desc, buffer) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(c_ptr) :: buffer
    integer(c_int) :: res
    res = cudecompFreeC(handle, grid_desc, buffer)
    buffer = c_null_ptr
  end function cudecompFreePtr

  ! ---- transposes -----------------------------------------------------------------------------------------------
  ! input/output/work: any array (or first element of one) whose address is a device address; type, kind and
  ! rank are not checked, `dtype` says what the elements are.  input and output may be the same array
  ! (in-place).
  function cudecompTransposeXToY(handle, grid_desc, input, output, work, dtype, input_halo_extents, &
                                 output_halo_extents, input_padding, output_padding, stream) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(*), dimension(..), target :: input, output, work
    integer :: dtype
    integer, optional :: input_halo_extents(3), output_halo_extents(3), input_padding(3), output_padding(3)
    integer(cudecomp_stream_kind), optional :: stream
    integer(c_int) :: res
    integer(c_int32_t) :: ih(3), oh(3), ip(3), op(3)
    integer(c_intptr_t) :: s
    call transpose_defaults(ih, oh, ip, op, s, input_halo_extents, output_halo_extents, input_padding, &
                            output_padding, stream)
    res = cudecompTransposeXToY_C(handle, grid_desc, c_loc(input), c_loc(output), c_loc(work), int(dtype, c_int), &
                                  ih, oh, ip, op, s)
  end function cudecompTransposeXToY

  function cudecompTransposeYToZ(handle, grid_desc, input, output, work, dtype, input_halo_extents, &
                                 output_halo_extents, input_padding, output_padding, stream) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(*), dimension(..), target :: input, output, work
    integer :: dtype
    integer, optional :: input_halo_extents(3), output_halo_extents(3), input_padding(3), output_padding(3)
    integer(cudecomp_stream_kind), optional :: stream
    integer(c_int) :: res
    integer(c_int32_t) :: ih(3), oh(3), ip(3), op(3)
    integer(c_intptr_t) :: s
    call transpose_defaults(ih, oh, ip, op, s, input_halo_extents, output_halo_extents, input_padding, &
                            output_padding, stream)
    res = cudecompTransposeYToZ_C(handle, grid_desc, c_loc(input), c_loc(output), c_loc(work), int(dtype, c_int), &
                                  ih, oh, ip, op, s)
  end function cudecompTransposeYToZ

  function cudecompTransposeZToY(handle, grid_desc, input, output, work, dtype, input_halo_extents, &
                                 output_halo_extents, input_padding, output_padding, stream) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(*), dimension(..), target :: input, output, work
    integer :: dtype
    integer, optional :: input_halo_extents(3), output_halo_extents(3), input_padding(3), output_padding(3)
    integer(cudecomp_stream_kind), optional :: stream
    integer(c_int) :: res
    integer(c_int32_t) :: ih(3), oh(3), ip(3), op(3)
    integer(c_intptr_t) :: s
    call transpose_defaults(ih, oh, ip, op, s, input_halo_extents, output_halo_extents, input_padding, &
                            output_padding, stream)
    res = cudecompTransposeZToY_C(handle, grid_desc, c_loc(input), c_loc(output), c_loc(work), int(dtype, c_int), &
                                  ih, oh, ip, op, s)
  end function cudecompTransposeZToY

  function cudecompTransposeYToX(handle, grid_desc, input, output, work, dtype, input_halo_extents, &
                                 output_halo_extents, input_padding, output_padding, stream) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(*), dimension(..), target :: input, output, work
    integer :: dtype
    integer, optional :: input_halo_extents(3), output_halo_extents(3), input_padding(3), output_padding(3)
    integer(cudecomp_stream_kind), optional :: stream
    integer(c_int) :: res
    integer(c_int32_t) :: ih(3), oh(3), ip(3), op(3)
    integer(c_intptr_t) :: s
    call transpose_defaults(ih, oh, ip, op, s, input_halo_extents, output_halo_extents, input_padding, &
                            output_padding, stream)
    res = cudecompTransposeYToX_C(handle, grid_desc, c_loc(input), c_loc(output), c_loc(work), int(dtype, c_int), &
                                  ih, oh, ip, op, s)
  end function cudecompTransposeYToX

  subroutine transpose_defaults(ih, oh, ip, op, s, input_halo_extents, output_halo_extents, input_padding, &
                                output_padding, stream)
    integer(c_int32_t), intent(out) :: ih(3), oh(3), ip(3), op(3)
    integer(c_intptr_t), intent(out) :: s
    integer, optional, intent(in) :: input_halo_extents(3), output_halo_extents(3), input_padding(3), &
                                     output_padding(3)
    integer(cudecomp_stream_kind), optional, intent(in) :: stream
    ih = 0
    oh = 0
    ip = 0
    op = 0
    s = 0
    if (present(input_halo_extents)) ih = int(input_halo_extents, c_int32_t)
    if (present(output_halo_extents)) oh = int(output_halo_extents, c_int32_t)
    if (present(input_padding)) ip = int(input_padding, c_int32_t)
    if (present(output_padding)) op = int(output_padding, c_int32_t)
    if (present(stream)) s = stream
  end subroutine transpose_defaults

  ! ---- halo updates ---------------------------------------------------------------------------------------------
  function cudecompUpdateHalosX(handle, grid_desc, input, work, dtype, halo_extents, halo_periods, dim, padding, &
                                stream) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(*), dimension(..), target :: input, work
    integer :: dtype
    integer :: halo_extents(3)
    logical :: halo_periods(3)
    integer :: dim  ! 1/2/3 = x/y/z
    integer, optional :: padding(3)
    integer(cudecomp_stream_kind), optional :: stream
    integer(c_int) :: res
    integer(c_int32_t) :: h(3), p(3)
    logical(c_bool) :: per(3)
    integer(c_intptr_t) :: s
    call halo_defaults(h, per, p, s, halo_extents, halo_periods, padding, stream)
    res = cudecompUpdateHalosX_C(handle, grid_desc, c_loc(input), c_loc(work), int(dtype, c_int), h, per, &
                                 int(dim - 1, c_int32_t), p, s)
  end function cudecompUpdateHalosX

  function cudecompUpdateHalosY(handle, grid_desc, input, work, dtype, halo_extents, halo_periods, dim, padding, &
                                stream) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(*), dimension(..), target :: input, work
    integer :: dtype
    integer :: halo_extents(3)
    logical :: halo_periods(3)
    integer :: dim
    integer, optional :: padding(3)
    integer(cudecomp_stream_kind), optional :: stream
    integer(c_int) :: res
    integer(c_int32_t) :: h(3), p(3)
    logical(c_bool) :: per(3)
    integer(c_intptr_t) :: s
    call halo_defaults(h, per, p, s, halo_extents, halo_periods, padding, stream)
    res = cudecompUpdateHalosY_C(handle, grid_desc, c_loc(input), c_loc(work), int(dtype, c_int), h, per, &
                                 int(dim - 1, c_int32_t), p, s)
  end function cudecompUpdateHalosY

  function cudecompUpdateHalosZ(handle, grid_desc, input, work, dtype, halo_extents, halo_periods, dim, padding, &
                                stream) result(res)
    type(cudecompHandle) :: handle
    type(cudecompGridDesc) :: grid_desc
    type(*), dimension(..), target :: input, work
    integer :: dtype
    integer :: halo_extents(3)
    logical :: halo_periods(3)
    integer :: dim
    integer, optional :: padding(3)
    integer(cudecomp_stream_kind), optional :: stream
    integer(c_int) :: res
    integer(c_int32_t) :: h(3), p(3)
    logical(c_bool) :: per(3)
    integer(c_intptr_t) :: s
    call halo_defaults(h, per, p, s, halo_extents, halo_periods, padding, stream)
    res = cudecompUpdateHalosZ_C(handle, grid_desc, c_loc(input), c_loc(work), int(dtype, c_int), h, per, &
                                 int(dim - 1, c_int32_t), p, s)
  end function cudecompUpdateHalosZ

  subroutine halo_defaults(h, per, p, s, halo_extents, halo_periods, padding, stream)
    integer(c_int32_t), intent(out) :: h(3), p(3)
    logical(c_bool), intent(out) :: per(3)
    integer(c_intptr_t), intent(out) :: s
    integer, intent(in) :: halo_extents(3)
    logical, intent(in) :: halo_periods(3)
    integer, optional, intent(in) :: padding(3)
    integer(cudecomp_stream_kind), optional, intent(in) :: stream
    h = int(halo_extents, c_int32_t)
    per = halo_periods
    p = 0
    s = 0
    if (present(padding)) p = int(padding, c_int32_t)
    if (present(stream)) s = stream
  end subroutine halo_defaults

end module cudecomp
