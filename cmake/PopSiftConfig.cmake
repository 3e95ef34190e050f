# Build-tree package: find_package(PopSift CONFIG) with -DPopSift_DIR=<repo>/cmake imports the libraries that
# popsift_amd/build.py (or __graft_entry__.build()) has produced in popsift_amd/lib, under the same target name a
# consumer of the reference uses: PopSift::popsift (README.md:60-71 of the reference).
get_filename_component(_popsift_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(_popsift_lib "${_popsift_root}/popsift_amd/lib/libpopsift.so")
set(_popsift_hip "${_popsift_root}/popsift_amd/lib/libpopsift_hip.so")
if(NOT EXISTS "${_popsift_lib}" OR NOT EXISTS "${_popsift_hip}")
    set(PopSift_FOUND FALSE)
    set(PopSift_NOT_FOUND_MESSAGE "popsift_amd/lib/libpopsift.so is missing: run `python -m popsift_amd.build` first")
    return()
endif()
include(CMakeFindDependencyMacro)
find_dependency(Threads)
if(NOT TARGET PopSift::popsift)
    add_library(PopSift::popsift SHARED IMPORTED)
    set_target_properties(PopSift::popsift PROPERTIES
        IMPORTED_LOCATION "${_popsift_lib}"
        IMPORTED_NO_SONAME ON
        INTERFACE_INCLUDE_DIRECTORIES "${_popsift_root}/popsift_amd/csrc/include;${_popsift_root}/include"
        INTERFACE_LINK_LIBRARIES "${_popsift_hip};Threads::Threads")
endif()
set(PopSift_VERSION 1.0.0)
set(PopSift_FOUND TRUE)
