# Build of the MI355X path-tracing hot path.
#   make            host library + HIP library (gfx950) + oracle (+ oracle/_ref when the reference is present)
#   make host       libtrc_host.so      C++17, CPU only
#   make hip        libtracer_amd.so    hand-written HIP for gfx950 (hipcc cross-compiles without a GPU)
#   make hip_hooks  libtracer_amd_hooks.so  the same sources + the test hooks of include/tracer_test_hooks.h (tests/, tools/)
#   make oracle     oracle/liboracle.so (test infrastructure, see oracle/README.md)
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CXX       ?= g++
LIBDIR    := tracer_amd/lib

# No FMA contraction anywhere: kernel and oracle must round every operation identically.
CXXFLAGS  := -std=c++17 -O2 -fPIC -Wall -Wextra -ffp-contract=off -Iinclude
# -fno-slp-vectorize: the SLP vectoriser packs pairs of adjacent scalar fp32 adds / muls into v_pk_add_f32 / v_pk_mul_f32
# and pays for it with v_mov shuffles in this branchy scalar code (347 packed ops in k_render): without it config 2 runs
# 22.7 -> 21.5 ms, config 3 70.1 -> 63.2 ms, config 4 35.3 -> 34.8 ms; the results are the same bits either way.
HIPFLAGS  := -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Iinclude -Itracer_amd/csrc \
             -Wall -Wno-unused-function

HOST_SRC  := tracer_amd/host/bvh_builder.cpp tracer_amd/host/scene.cpp tracer_amd/host/mesh.cpp tracer_amd/host/pbrt_scene.cpp
HOST_HDR  := tracer_amd/host/host_math.hpp tracer_amd/host/host_scene.hpp tracer_amd/host/pbrt_text.hpp include/tracer_abi.h include/trc_sobol.h
HIP_SRC   := tracer_amd/csrc/trc_abi.hip tracer_amd/csrc/trc_render_lds.hip tracer_amd/csrc/trc_render_lds_mis.hip tracer_amd/csrc/trc_render_mem.hip tracer_amd/csrc/trc_render_mem_path.hip tracer_amd/csrc/trc_render_mem_volume.hip tracer_amd/csrc/trc_sppm.hip tracer_amd/csrc/trc_lbvh.hip
# per translation unit: backend options that pay for ONE kernel family (profiles/r05/ab_flags*.txt: eight scheduler / sinking / LICM options
# tried on configs 2 / 3 / 4; everything else is within +-1 % or worse).  Scheduling only: the parity suites run on this build.
EXTRA_trc_render_mem_path := -mllvm -disable-machine-sink
EXTRA_trc_render_mem_volume := -mllvm -disable-machine-sink
EXTRA_trc_render_lds      := -mllvm -amdgpu-use-amdgpu-trackers
HIP_HDR   := $(wildcard tracer_amd/csrc/*.hpp) include/tracer_abi.h include/tracer_test_hooks.h include/trc_detmath.h include/trc_sobol.h

.PHONY: all host hip hip_fast hip_hooks oracle example clean variant asan tsan sanitize design_table
all: host hip hip_fast hip_hooks oracle example

host: $(LIBDIR)/libtrc_host.so
hip: $(LIBDIR)/libtracer_amd.so
hip_fast: $(LIBDIR)/libtracer_amd_fast.so
hip_hooks: $(LIBDIR)/libtracer_amd_hooks.so
oracle:
	$(MAKE) -C oracle

$(LIBDIR)/libtrc_host.so: $(HOST_SRC) $(HOST_HDR) Makefile
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -lpthread

# One object per translation unit (make -j compiles them side by side; `make` with no -j still works), then one link.
# RCCL is resolved at run time (dlopen in trc_group_*), so the library loads on boxes without it.
HIP_OBJ      := $(patsubst tracer_amd/csrc/%.hip,build/obj/exact/%.o,$(HIP_SRC))
HIP_OBJ_FAST := $(patsubst tracer_amd/csrc/%.hip,build/obj/fast/%.o,$(HIP_SRC))
build/obj/exact/%.o: tracer_amd/csrc/%.hip $(HIP_HDR) Makefile
	@mkdir -p build/obj/exact
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) -c -o $@ $<
$(LIBDIR)/libtracer_amd.so: $(HIP_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -o $@ $(HIP_OBJ) -ldl -lpthread -Wl,-rpath,$(ROCM)/lib

# Product and laboratory apart: the entry points of include/tracer_test_hooks.h (exhaustive arithmetic checks, the SPPM hash,
# the per-site cycle profile) exist only in this build of the SAME sources.  Only the two translation units that define them are
# compiled again; the render kernels are the product's objects, byte for byte.
HOOK_TU      := trc_abi trc_sppm
HIP_OBJ_HOOKS := $(foreach o,$(HIP_OBJ),$(if $(filter $(HOOK_TU),$(basename $(notdir $(o)))),build/obj/hooks/$(notdir $(o)),$(o)))
build/obj/hooks/%.o: tracer_amd/csrc/%.hip $(HIP_HDR) Makefile
	@mkdir -p build/obj/hooks
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) -DTRC_TEST_HOOKS=1 -c -o $@ $<
$(LIBDIR)/libtracer_amd_hooks.so: $(HIP_OBJ_HOOKS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -o $@ $(HIP_OBJ_HOOKS) -ldl -lpthread -Wl,-rpath,$(ROCM)/lib

# The same sources under fast-math rules (what the reference's shaders are compiled with: MTL_FAST_MATH): approximate
# division / sqrt (v_rcp_f32, v_sqrt_f32), FMA contraction, denormals flushed, hardware exp / log / sin / cos
# (include/trc_detmath.h under TRC_FAST_MATH).  NaN / Inf semantics and signed zeros are
# kept (the integrators scrub NaN samples, Render.metal:537-538).  NOT comparable bit for bit with the oracle: parity of
# this build is statistical (tests/test_gpu_fast_math.py); trc_build_flavor() tells a host which one it loaded.
FASTFLAGS := -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast -fgpu-flush-denormals-to-zero -DTRC_FAST_MATH=1
build/obj/fast/%.o: tracer_amd/csrc/%.hip $(HIP_HDR) Makefile
	@mkdir -p build/obj/fast
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) $(FASTFLAGS) -c -o $@ $<
$(LIBDIR)/libtracer_amd_fast.so: $(HIP_OBJ_FAST)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -o $@ $(HIP_OBJ_FAST) -ldl -lpthread -Wl,-rpath,$(ROCM)/lib

# A/B variants of the device library for tools/ab_bench.py:  make -j variant NAME=pwg12 DEFS=-DTRC_PWG_WAVES_PATH=12
# The object directory is keyed on the definitions as well as the name: the same NAME with other DEFS compiles afresh instead of
# relinking the objects of the earlier definitions (an A/B table must never measure a stale build).
VDIR := build/obj/v_$(NAME)_$(shell printf '%s' '$(DEFS)' | md5sum | cut -c1-8)
$(VDIR)/%.o: tracer_amd/csrc/%.hip $(HIP_HDR) Makefile
	@mkdir -p $(VDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) $(DEFS) -c -o $@ $<
variant: $(patsubst tracer_amd/csrc/%.hip,$(VDIR)/%.o,$(HIP_SRC))
	$(HIPCC) --offload-arch=gfx950 -shared -o build/lib$(NAME).so $^ -ldl -lpthread -Wl,-rpath,$(ROCM)/lib

# C++ host driving the path through the C ABI only (no Python): examples/trc_render
example: examples/trc_render examples/trc_ranks
# N ranks started and composed without Python / PyTorch: RCCL id or the collectives themselves over TCP sockets
examples/trc_ranks: examples/trc_ranks.cpp include/tracer_abi.h $(LIBDIR)/libtrc_host.so $(LIBDIR)/libtracer_amd.so
	$(CXX) -std=c++17 -O2 -Wall -Iinclude -o $@ examples/trc_ranks.cpp -L$(LIBDIR) -ltracer_amd -ltrc_host -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,$(ROCM)/lib
examples/trc_render: examples/trc_render.cpp include/tracer_abi.h $(LIBDIR)/libtrc_host.so $(LIBDIR)/libtracer_amd.so
	$(CXX) -std=c++17 -O2 -Wall -Iinclude -o $@ examples/trc_render.cpp -L$(LIBDIR) -ltracer_amd -ltrc_host -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,$(ROCM)/lib

# Sanitizers: the CPU sources of the repository -- libtrc_host and the oracle -- compiled with the
# sanitizer into one driver (tools/sanitize/driver.cpp: threaded SAH builds, the oracle's row-band workers, every file reader
# on well-formed and on mutated files), plus sanitized shared libraries for the Python CPU suite (tools/run_sanitizers.sh
# preloads the runtime).  GPU code is not covered: GPU AddressSanitizer is not available on this pool.
SAN_SRC := tools/sanitize/driver.cpp $(HOST_SRC) oracle/oracle.cpp oracle/oracle_lbvh.cpp oracle/oracle_sah.cpp
build/asan/driver: $(SAN_SRC) $(HOST_HDR) oracle/oracle.h
	@mkdir -p build/asan
	$(CXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=undefined,float-cast-overflow -ffp-contract=off -Iinclude -o $@ $(SAN_SRC) -lpthread
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=address,undefined,float-cast-overflow -shared -o build/asan/libtrc_host.so $(HOST_SRC) -lpthread
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=address,undefined -Wno-unused-function -shared -o build/asan/liboracle.so oracle/oracle.cpp oracle/oracle_lbvh.cpp oracle/oracle_sah.cpp -lpthread
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=address,undefined -Wno-unused-function -DORACLE_USE_LIBM -shared -o build/asan/liboracle_libm.so oracle/oracle.cpp oracle/oracle_lbvh.cpp oracle/oracle_sah.cpp -lpthread
build/tsan/driver: $(SAN_SRC) $(HOST_HDR) oracle/oracle.h
	@mkdir -p build/tsan
	$(CXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=thread -ffp-contract=off -Iinclude -o $@ $(SAN_SRC) -lpthread
asan: build/asan/driver
tsan: build/tsan/driver
sanitize:
	bash tools/run_sanitizers.sh

clean:
	rm -f $(LIBDIR)/*.so examples/trc_render examples/trc_ranks
	$(MAKE) -C oracle clean

design_table:
	python3 tools/design_table.py --write
