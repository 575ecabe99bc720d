# Builds the front-end library for MI355X (gfx950) and the host-only helpers.
#   vi_slam_amd/lib/libvfe.so        HIP kernels + C-ABI (include/vfe.h)
#   vi_slam_amd/lib/libvfe_synth.so  seeded synthetic inputs (host only)
#   oracle/libvfe_oracle.so          CPU oracle (test infrastructure only)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
LIBDIR = vi_slam_amd/lib
CSRC = vi_slam_amd/csrc
# -ffp-contract=off: every float operation rounds once, exactly as in the CPU oracle.
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$(CSRC) -Wall

HIP_SRCS = $(wildcard $(CSRC)/*.hip)
HIP_HDRS = $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) include/vfe.h

all: $(LIBDIR)/libvfe_synth.so $(LIBDIR)/libvfe.so oracle

$(LIBDIR):
	mkdir -p $(LIBDIR)

$(LIBDIR)/libvfe_synth.so: $(CSRC)/synth.cpp | $(LIBDIR)
	$(CXX) -O2 -std=c++17 -fPIC -shared -o $@ $<

$(LIBDIR)/libvfe.so: $(HIP_SRCS) $(HIP_HDRS) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS)

oracle:
	$(MAKE) -C oracle libvfe_oracle.so

oracle-fast:
	$(MAKE) -C oracle libvfe_oracle_fast.so

clean:
	rm -f $(LIBDIR)/*.so
	$(MAKE) -C oracle clean

.PHONY: all oracle oracle-fast clean
