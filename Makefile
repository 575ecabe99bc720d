# Builds the front-end library for MI355X (gfx950) and the host-only helpers.
#   vi_slam_amd/lib/libvfe.so        HIP kernels + C-ABI (include/vfe.h)
#   vi_slam_amd/lib/libvfe_synth.so  seeded synthetic inputs (host only)
#   oracle/libvfe_oracle.so          CPU oracle (test infrastructure only)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
LIBDIR ?= vi_slam_amd/lib
CSRC = vi_slam_amd/csrc
# -ffp-contract=off: every float operation rounds once, exactly as in the CPU oracle.
# EXTRA: experiment switches (-D...), used with LIBDIR=/OBJDIR= overrides to build a variant library beside the default one
EXTRA ?=
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$(CSRC) -Wall $(EXTRA)

HIP_SRCS = $(wildcard $(CSRC)/*.hip)
HIP_HDRS = $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) include/vfe.h

HOST = vi_slam_amd/host
HOST_SRCS = $(wildcard $(HOST)/src/*.cpp)
HOST_HDRS = $(wildcard $(HOST)/include/*/*.hpp) $(wildcard $(HOST)/include/*/*/*.hpp) include/vfe.h
CPP_TESTS = $(patsubst tests/cpp/%.cpp,tests/cpp/bin/%,$(wildcard tests/cpp/*.cpp)) tests/cpp/bin/pc_client_video_callback

HIP_TESTS = $(patsubst tests/hip/%.hip,tests/cpp/bin/%,$(wildcard tests/hip/*.hip)) tests/cpp/bin/valu_ceiling tests/cpp/bin/hbm_stream tests/cpp/bin/mfma_ceiling

all: $(LIBDIR)/libvfe_synth.so $(LIBDIR)/libvfe.so $(LIBDIR)/libvi_slam_amd.so $(CPP_TESTS) $(HIP_TESTS) oracle

$(LIBDIR):
	mkdir -p $(LIBDIR)

$(LIBDIR)/libvfe_synth.so: $(CSRC)/synth.cpp | $(LIBDIR)
	$(CXX) -O2 -std=c++17 -fPIC -shared -o $@ $<

# one object per source so that `make -j` compiles them side by side (no device code is shared across files)
OBJDIR ?= build/obj
HIP_OBJS = $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(HIP_SRCS))
# the eigenvalue kernel chooses between packed and plain float operations by hand (issue cost per element is the same, the
# packed form needs its operands in register pairs); the SLP vectoriser would re-pack the plain ones behind extra moves
$(OBJDIR)/vfe_gftt_eig.o: HIPFLAGS += -fno-slp-vectorize
$(OBJDIR)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIBDIR)/libvfe.so: $(HIP_OBJS) | $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJS)

# host side: SLAMEngine + framework adapters (C++17), linked against libvfe.so next to it
$(LIBDIR)/libvi_slam_amd.so: $(HOST_SRCS) $(HOST_HDRS) $(LIBDIR)/libvfe.so
	$(CXX) -O2 -std=c++17 -fPIC -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$(HOST)/include -Iinclude -shared -o $@ $(HOST_SRCS) -L$(LIBDIR) -lvfe -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -lpthread

# the soak program also asks the HIP runtime how much device memory is in use
tests/cpp/bin/test_soak: tests/cpp/test_soak.cpp tests/cpp/gpu_test_frames.hpp $(LIBDIR)/libvi_slam_amd.so
	@mkdir -p tests/cpp/bin
	$(CXX) -O1 -std=c++17 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$(HOST)/include -Iinclude -o $@ $< -L$(LIBDIR) -lvi_slam_amd -lvfe -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../../../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib -lpthread

# lifecycle tests in the reference's style (plain main(), exit code 1 on failure)
tests/cpp/bin/%: tests/cpp/%.cpp tests/cpp/gpu_test_frames.hpp $(LIBDIR)/libvi_slam_amd.so
	@mkdir -p tests/cpp/bin
	$(CXX) -O1 -std=c++17 -Wall -I$(HOST)/include -Iinclude -o $@ $< -L$(LIBDIR) -lvi_slam_amd -lvfe -Wl,-rpath,'$$ORIGIN/../../../$(LIBDIR)' -lpthread

# the PC client's receiver callbacks written out (compiles against the host layer only)
tests/cpp/bin/pc_client_video_callback: $(HOST)/examples/pc_client_video_callback.cpp $(LIBDIR)/libvi_slam_amd.so
	@mkdir -p tests/cpp/bin
	$(CXX) -O1 -std=c++17 -Wall -I$(HOST)/include -Iinclude -o $@ $< -L$(LIBDIR) -lvi_slam_amd -lvfe -Wl,-rpath,'$$ORIGIN/../../../$(LIBDIR)' -lpthread

# standalone device checks (plain main(), exit code 1 on a mismatch)
tests/cpp/bin/%: tests/hip/%.hip $(HIP_HDRS)
	@mkdir -p tests/cpp/bin
	$(HIPCC) $(HIPFLAGS) -o $@ $<

# measurement tool: VALU issue ceiling per instruction class (tools/valu_ceiling.hip)
tests/cpp/bin/valu_ceiling: tools/valu_ceiling.hip
	@mkdir -p tests/cpp/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -o $@ $<

# measurement tool: what the matrix pipe sustains on the matcher's instruction (tools/mfma_ceiling.hip)
tests/cpp/bin/mfma_ceiling: tools/mfma_ceiling.hip
	@mkdir -p tests/cpp/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -o $@ $<

# measurement tool: what HBM delivers to plain streaming kernels (tools/hbm_stream.hip)
tests/cpp/bin/hbm_stream: tools/hbm_stream.hip
	@mkdir -p tests/cpp/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -o $@ $<

oracle:
	$(MAKE) -C oracle libvfe_oracle.so

oracle-fast:
	$(MAKE) -C oracle libvfe_oracle_fast.so

clean:
	rm -f $(LIBDIR)/*.so
	rm -rf tests/cpp/bin
	$(MAKE) -C oracle clean

.PHONY: all oracle oracle-fast clean
