# owl_mi355x.cmake -- drop-in for the reference's owl/cmake/configure_owl.cmake +
# configure_optix.cmake as far as samples need them:
#   OWL_INCLUDES, OWL_LIBRARIES, and the macro cuda_compile_and_embed(<var> <file.cu>), which here
#   compiles the device file for gfx950 with hipcc and embeds the code object (tools/owl_embed.py).
# A sample's CMakeLists.txt (e.g. the reference's samples/s01-trueknn/CMakeLists.txt:19-30) works
# unchanged after  include(<repo>/cmake/owl_mi355x.cmake).
get_filename_component(OWL_MI355X_ROOT "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(OWL_INCLUDES ${OWL_MI355X_ROOT}/include ${OWL_MI355X_ROOT}/include/owl_shims /opt/rocm/include)
set(OWL_LIBRARIES ${OWL_MI355X_ROOT}/owlraytracing_amd/libowl_mi355x.so /opt/rocm/lib/libamdhip64.so)
include_directories(${OWL_INCLUDES})
add_definitions(-D__HIP_PLATFORM_AMD__=1)
find_package(Python3 REQUIRED COMPONENTS Interpreter)

macro(cuda_compile_and_embed output_var cuda_file)
  set(_owl_c ${CMAKE_CURRENT_BINARY_DIR}/${output_var}.c)
  get_filename_component(_owl_src ${cuda_file} ABSOLUTE)
  get_directory_property(_owl_dirs INCLUDE_DIRECTORIES)
  set(_owl_inc)
  foreach(d ${_owl_dirs})
    list(APPEND _owl_inc -I ${d})
  endforeach()
  add_custom_command(
    OUTPUT ${_owl_c}
    COMMAND ${Python3_EXECUTABLE} ${OWL_MI355X_ROOT}/tools/owl_embed.py ${output_var} ${_owl_src} -o ${_owl_c}
            -I ${CMAKE_CURRENT_SOURCE_DIR} ${_owl_inc}
    DEPENDS ${_owl_src}
    COMMENT "hipcc (gfx950) + embed: ${cuda_file} -> ${output_var}")
  set(${output_var} ${_owl_c})
endmacro()
