# weldacs_dropin.cmake -- make mhsitu/welding_robot's main.cpp consume libweldacs through the drop-in headers.
#
# In the reference's top-level CMakeLists.txt, AFTER its include_directories(...) block and INSTEAD OF
#     file(GLOB USER_SOURCE "*.cpp")
#     add_executable(${PROJECT_NAME} ${USER_SOURCE})
# write
#     set(WELDACS /path/to/this/repo)
#     include(${WELDACS}/welding_robot_amd/cmake/weldacs_dropin.cmake)
#
# Why a shadow copy: main.cpp includes the planning headers with QUOTES ("core/ACSRank_3D.hpp", main.cpp:6-11).
# A quoted include is looked up in the including file's own directory first, so while main.cpp sits next to the
# reference's core/, no -I order can shadow core/*.hpp -- the old CPU headers would be compiled and libweldacs.so
# linked but never called.  Compiling an unmodified COPY of main.cpp from the build tree removes that first hit;
# the five planning headers then resolve to ${WELDACS}/welding_robot_amd/include/core/ (searched BEFORE the
# reference root), every other header ("core/Timer.h", "core/BezierCurve.h", common/, coppeliaSim-client/) still
# resolves to the reference.  tests/test_consumer_compile.py checks exactly this resolution with `g++ -H`.
if(NOT WELDACS)
  message(FATAL_ERROR "set(WELDACS <path to the weldacs repo>) before including weldacs_dropin.cmake")
endif()
set(WELDACS_SHADOW ${CMAKE_BINARY_DIR}/weldacs_shadow)
file(GLOB WELDACS_USER_SOURCE "${CMAKE_SOURCE_DIR}/*.cpp")
set(WELDACS_SHADOW_SOURCE "")
foreach(src ${WELDACS_USER_SOURCE})
  get_filename_component(name ${src} NAME)
  configure_file(${src} ${WELDACS_SHADOW}/${name} COPYONLY)   # re-copied whenever the original changes
  list(APPEND WELDACS_SHADOW_SOURCE ${WELDACS_SHADOW}/${name})
endforeach()
include_directories(BEFORE ${WELDACS}/include ${WELDACS}/welding_robot_amd/include)   # weldacs.h, core/*.hpp
add_executable(${PROJECT_NAME} ${WELDACS_SHADOW_SOURCE})
# the drop-in ACS_Rank runs one host thread per device shard (std::thread)
find_package(Threads REQUIRED)
target_link_libraries(${PROJECT_NAME} ${WELDACS}/welding_robot_amd/lib/libweldacs.so Threads::Threads)
# main.cpp itself still includes matplotlibcpp.h (main.cpp:2), so Python3/NumPy stay for the application shell;
# the planning headers no longer need them (define WELDACS_WITH_MATPLOTLIB to keep plot_grid_map()/plot_path()).
