# Top-level build.  Everything is built in-tree so that the .so files travel to
# the GPU box with the snapshot (they are git-ignored, not gpurun-ignored).
#
#   make            product libraries + tools + oracle (+ oracle/_ref when the
#                   upstream sources are present) + the CPU test shim
#   make product    only the product: libfpvhip.so (HIP, gfx950) and
#                   libfpv_amd.so (host codec over the C ABI)

PKG      := fusion-power-video_amd
CSRC     := $(PKG)/csrc
LIBDIR   := $(PKG)/lib
BINDIR   := build/bin

HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
ROCM_LIB ?= /opt/rocm/lib
BROTLI_I ?= /opt/conda/include
BROTLI_L ?= /usr/lib/x86_64-linux-gnu
BROTLI   := $(BROTLI_L)/libbrotlienc.so.1 $(BROTLI_L)/libbrotlidec.so.1 $(BROTLI_L)/libbrotlicommon.so.1
REF      ?= /root/reference

HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-value -Wno-unused-result
CXXFLAGS := -std=c++14 -O2 -fPIC -Iinclude -I$(CSRC) -I$(BROTLI_I) -Wall -Wextra -pthread

HOST_SRC := $(CSRC)/fpv_host.cc $(CSRC)/fpv_decoders.cc $(CSRC)/fpv_frame.cc $(CSRC)/fpv_c_api.cc $(CSRC)/fpv_columnar.cc
HOST_HDR := include/fusion_power_video.h include/fpv_hip.h include/fpv_c_api.h $(CSRC)/fpv_host_internal.h $(CSRC)/fpv_frame_capi.inc \
            $(wildcard include/columnar_batch/*.h)

all: product tools oracle shim refcli refcb columnar arrow

product: $(LIBDIR)/libfpvhip.so $(LIBDIR)/libfpv_amd.so

$(LIBDIR)/libfpvhip.so: $(CSRC)/fpv_kernels.hip $(CSRC)/fpv_tile.inc $(CSRC)/fpv_tile8.inc $(CSRC)/fpv_inverse.inc $(CSRC)/fpv_inv_pass.inc $(CSRC)/fpv_capi.hip $(CSRC)/fpv_kernels.h include/fpv_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/fpv_kernels.hip $(CSRC)/fpv_capi.hip \
	    -Wl,-rpath,$(ROCM_LIB) -o $@

# Host codec: g++ only, calls the kernels through the C ABI of libfpvhip.so.
$(LIBDIR)/libfpv_amd.so: $(HOST_SRC) $(HOST_HDR) $(LIBDIR)/libfpvhip.so
	$(CXX) $(CXXFLAGS) -shared $(HOST_SRC) -L$(LIBDIR) -lfpvhip $(BROTLI) \
	    -Wl,-rpath,'$$ORIGIN' -o $@

# Our own CLI tools (written from scratch; usable on the GPU box) and the native test driver
# of the BASELINE configurations at their stated frame counts.
tools: $(BINDIR)/fpv_encode $(BINDIR)/fpv_decode $(BINDIR)/fpv_benchmark $(BINDIR)/configs_driver $(BINDIR)/frame_rate $(BINDIR)/cold_probe
# what any fresh process pays for its first GPU milliseconds (tests/test_gpu_cli.py: the floor of a cold start)
$(BINDIR)/cold_probe: tools/exp/cold_probe.hip
	@mkdir -p $(BINDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O2 -w $< -pthread -o $@
$(BINDIR)/configs_driver: tests/configs_driver.cc $(LIBDIR)/libfpv_amd.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) $< -L$(LIBDIR) -lfpv_amd -lfpvhip \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@
$(BINDIR)/fpv_%: tools/fpv_%.cc $(LIBDIR)/libfpv_amd.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) $< -L$(LIBDIR) -lfpv_amd -lfpvhip \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@

# The upstream CLI sources compiled UNCHANGED against our header + library
# (drop-in proof).  Only where the upstream tree is present.
ifneq ($(wildcard $(REF)/encode.cc),)
refcli: $(BINDIR)/ref_encode $(BINDIR)/ref_decode $(BINDIR)/ref_benchmark
# The sources are reached through symlinks in build/refsrc so that their
# `#include "fusion_power_video.h"` finds OUR header (-Iinclude) and not the
# upstream one sitting next to them.
.SECONDARY:
build/refsrc/%.cc: $(REF)/%.cc
	@mkdir -p build/refsrc
	ln -sf $< $@
$(BINDIR)/ref_%: build/refsrc/%.cc $(LIBDIR)/libfpv_amd.so include/fusion_power_video.h
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++14 -O2 -w -Iinclude $< -L$(LIBDIR) -lfpv_amd -lfpvhip -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@

# The upstream columnar_batch adapter (uses the Frame API) compiled unchanged:
# its sources say #include "../fusion_power_video.h", so a symlink to OUR
# header sits one level above the symlinked sources.
CB := columnar_batch
refcb: $(BINDIR)/ref_columnar_batch_decoder_test $(BINDIR)/ref_columnar_batch_encoder_test
build/refsrc/fusion_power_video.h: include/fusion_power_video.h
	@mkdir -p build/refsrc
	ln -sf ../../include/fusion_power_video.h $@
build/refsrc/$(CB)/.linked: $(wildcard $(REF)/$(CB)/*) build/refsrc/fusion_power_video.h
	@mkdir -p build/refsrc/$(CB)
	for f in $(REF)/$(CB)/*.cc $(REF)/$(CB)/*.h; do ln -sf $$f build/refsrc/$(CB)/; done
	touch $@
$(BINDIR)/ref_columnar_batch_%_test: build/refsrc/$(CB)/.linked $(LIBDIR)/libfpv_amd.so
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++17 -O2 -w -I$(BROTLI_I) build/refsrc/$(CB)/columnar_batch.cc \
	    build/refsrc/$(CB)/columnar_batch_encoder.cc build/refsrc/$(CB)/columnar_batch_decoder.cc \
	    build/refsrc/$(CB)/columnar_batch_$*_test.cc -L$(LIBDIR) -lfpv_amd -lfpvhip $(BROTLI) -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@
else
refcli:
	@echo "refcli: $(REF) not present, skipped"
refcb:
	@echo "refcb: $(REF) not present, skipped"
endif

# Native columnar batch adapter: our own test driver over the product library and over
# the CPU test shim, and (where the upstream tree is present) the upstream columnar
# tests compiled UNCHANGED against OUR columnar headers (symlinked so that their quoted
# includes resolve to include/columnar_batch/).
columnar: $(BINDIR)/columnar_driver tests/shim/columnar_driver_cpushim $(BINDIR)/columnar_dump tests/shim/columnar_dump_cpushim tests/shim/adapter_fail_cpushim
# GPU failures inside the adapters are latched, never fatal (tests/adapter_fail_driver.cc; CPU shim only: it injects them)
tests/shim/adapter_fail_cpushim: tests/adapter_fail_driver.cc tests/shim/libfpv_amd_cpushim.so $(wildcard include/columnar_batch/*.h)
	$(CXX) -std=c++14 -O2 -Wall -Iinclude/columnar_batch $< tests/shim/libfpv_amd_cpushim.so -lpthread \
	    -Wl,-rpath,'$$ORIGIN' -o $@
# format-parity dump of the native Batch (tests/columnar_dump.cc; the reference build of the same
# source lives in oracle/_ref)
$(BINDIR)/columnar_dump: tests/columnar_dump.cc $(LIBDIR)/libfpv_amd.so $(wildcard include/columnar_batch/*.h)
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++14 -O2 -Wall -Iinclude/columnar_batch $< -L$(LIBDIR) -lfpv_amd -lfpvhip -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@
tests/shim/columnar_dump_cpushim: tests/columnar_dump.cc tests/shim/libfpv_amd_cpushim.so $(wildcard include/columnar_batch/*.h)
	$(CXX) -std=c++14 -O2 -Wall -Iinclude/columnar_batch $< tests/shim/libfpv_amd_cpushim.so -lpthread \
	    -Wl,-rpath,'$$ORIGIN' -o $@
# Frame-API throughput (tests/frame_rate.cc; the reference build of the same source lives in oracle/_ref)
$(BINDIR)/frame_rate: tests/frame_rate.cc $(LIBDIR)/libfpv_amd.so include/fusion_power_video.h
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++14 -O2 -Wall -Iinclude $< -L$(LIBDIR) -lfpv_amd -lfpvhip -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@
$(BINDIR)/columnar_driver: tests/columnar_driver.cc $(LIBDIR)/libfpv_amd.so $(wildcard include/columnar_batch/*.h)
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++14 -O2 -Wall -Iinclude/columnar_batch $< -L$(LIBDIR) -lfpv_amd -lfpvhip -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@
tests/shim/columnar_driver_cpushim: tests/columnar_driver.cc tests/shim/libfpv_amd_cpushim.so $(wildcard include/columnar_batch/*.h)
	$(CXX) -std=c++14 -O2 -Wall -Iinclude/columnar_batch $< tests/shim/libfpv_amd_cpushim.so -lpthread \
	    -Wl,-rpath,'$$ORIGIN' -o $@
ifneq ($(wildcard $(REF)/columnar_batch/columnar_batch_decoder_test.cc),)
columnar: $(BINDIR)/ref_cbtest_native_decoder $(BINDIR)/ref_cbtest_native_encoder
build/refsrc_native/%.cc: $(REF)/columnar_batch/%.cc
	@mkdir -p build/refsrc_native
	ln -sf $< $@
$(BINDIR)/ref_cbtest_native_%: build/refsrc_native/columnar_batch_%_test.cc $(LIBDIR)/libfpv_amd.so $(wildcard include/columnar_batch/*.h)
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++17 -O2 -w -Iinclude/columnar_batch $< -L$(LIBDIR) -lfpv_amd -lfpvhip -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -o $@
endif

# Arrow record-batch encoder: needs an Arrow C++ installation.  The image
# has none of its own, but pyarrow ships the headers and libarrow; built only when found.
PYARROW := $(shell python3 -c "import pyarrow, sys; sys.stdout.write(pyarrow.get_library_dirs()[0])" 2>/dev/null)
ARROW_SO := $(firstword $(wildcard $(PYARROW)/libarrow.so.[0-9]*))
ifneq ($(ARROW_SO),)
ARROWFLAGS := -std=c++20 -O2 -fPIC -Wall -Iinclude -I$(CSRC) -I$(PYARROW)/include
arrow: $(LIBDIR)/libfpv_arrow.so $(BINDIR)/arrow_driver tests/shim/arrow_driver_cpushim $(BINDIR)/arrow_dump tests/shim/arrow_dump_cpushim
# format-parity dump of the native Arrow adapter (tests/arrow_dump.cc; the reference build of the
# same source lives in oracle/_ref)
$(BINDIR)/arrow_dump: tests/arrow_dump.cc $(LIBDIR)/libfpv_arrow.so
	@mkdir -p $(BINDIR)
	$(CXX) $(ARROWFLAGS) -Iinclude/fpv_arrow $< -L$(LIBDIR) -lfpv_arrow -lfpv_amd -lfpvhip $(ARROW_SO) -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -Wl,-rpath,$(PYARROW) -o $@
tests/shim/arrow_dump_cpushim: tests/arrow_dump.cc $(CSRC)/fpv_arrow.cc tests/shim/libfpv_amd_cpushim.so
	$(CXX) $(ARROWFLAGS) -Iinclude/fpv_arrow tests/arrow_dump.cc $(CSRC)/fpv_arrow.cc tests/shim/libfpv_amd_cpushim.so $(ARROW_SO) -lpthread \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(PYARROW) -o $@
$(LIBDIR)/libfpv_arrow.so: $(CSRC)/fpv_arrow.cc $(wildcard include/fpv_arrow/*.h) $(LIBDIR)/libfpv_amd.so
	$(CXX) $(ARROWFLAGS) -shared $< -L$(LIBDIR) -lfpv_amd $(ARROW_SO) -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(PYARROW) -o $@
$(BINDIR)/arrow_driver: tests/arrow_driver.cc $(LIBDIR)/libfpv_arrow.so
	@mkdir -p $(BINDIR)
	$(CXX) $(ARROWFLAGS) $< -L$(LIBDIR) -lfpv_arrow -lfpv_amd -lfpvhip $(ARROW_SO) -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -Wl,-rpath,$(PYARROW) -o $@
# CPU variant: the adapter's source together with the test shim build of the codec
tests/shim/arrow_driver_cpushim: tests/arrow_driver.cc $(CSRC)/fpv_arrow.cc tests/shim/libfpv_amd_cpushim.so
	$(CXX) $(ARROWFLAGS) tests/arrow_driver.cc $(CSRC)/fpv_arrow.cc tests/shim/libfpv_amd_cpushim.so $(ARROW_SO) -lpthread \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(PYARROW) -o $@
ifneq ($(wildcard $(REF)/arrow/arrow_encoder_test.cc),)
# upstream's Arrow test compiled UNCHANGED against OUR header + library, and upstream's own
# adapter compiled unchanged over our Frame API
arrow: $(BINDIR)/ref_arrowtest_native $(BINDIR)/ref_arrow_encoder_test
build/refsrc_native/arrow_encoder_test.cc: $(REF)/arrow/arrow_encoder_test.cc
	@mkdir -p build/refsrc_native
	ln -sf $< $@
$(BINDIR)/ref_arrowtest_native: build/refsrc_native/arrow_encoder_test.cc $(LIBDIR)/libfpv_arrow.so
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++20 -O2 -w -Iinclude/fpv_arrow -I$(PYARROW)/include $< -L$(LIBDIR) -lfpv_arrow -lfpv_amd -lfpvhip $(ARROW_SO) -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -Wl,-rpath,$(PYARROW) -o $@
build/refsrc/arrow/.linked: $(wildcard $(REF)/arrow/*) build/refsrc/fusion_power_video.h
	@mkdir -p build/refsrc/arrow
	for f in $(REF)/arrow/*.cc $(REF)/arrow/*.h; do ln -sf $$f build/refsrc/arrow/; done
	touch $@
$(BINDIR)/ref_arrow_encoder_test: build/refsrc/arrow/.linked $(LIBDIR)/libfpv_amd.so
	@mkdir -p $(BINDIR)
	$(CXX) -std=c++20 -O2 -w -I$(PYARROW)/include build/refsrc/arrow/arrow_encoder.cc build/refsrc/arrow/arrow_encoder_test.cc \
	    -L$(LIBDIR) -lfpv_amd -lfpvhip $(ARROW_SO) -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM_LIB) -Wl,-rpath,$(PYARROW) -o $@
endif
else
arrow:
	@echo "arrow: no Arrow C++ found (pyarrow not importable), skipped"
endif

oracle:
	$(MAKE) -C oracle REF=$(REF)

# TEST-ONLY: the host codec linked against an oracle-backed stand-in for the C
# ABI, so that the container / pipeline logic can be tested without a GPU.
# Lives under tests/, is never loaded by the product.
shim: tests/shim/libfpv_amd_cpushim.so
tests/shim/fpvhip_oracle_shim.o: tests/shim/fpvhip_oracle_shim.c include/fpv_hip.h oracle/fpv_oracle.h
	gcc -std=c99 -O2 -fPIC -Wall -Wextra -Iinclude -Ioracle -c $< -o $@
tests/shim/fpv_oracle.o: oracle/fpv_oracle.c oracle/fpv_oracle.h
	gcc -std=c99 -O2 -fPIC -Wall -Wextra -Ioracle -I$(BROTLI_I) -c $< -o $@
tests/shim/libfpv_amd_cpushim.so: tests/shim/fpvhip_oracle_shim.o tests/shim/fpv_oracle.o $(HOST_SRC) $(HOST_HDR)
	$(CXX) $(CXXFLAGS) -shared -Wl,-Bsymbolic $(HOST_SRC) tests/shim/fpvhip_oracle_shim.o tests/shim/fpv_oracle.o \
	    $(BROTLI) -o $@

# AddressSanitizer + UBSan build of the host codec over the CPU shim (sanitizers
# are CPU-only on this pool): tools/asan_host_tests.sh runs the host tests on it.
asan: build/asan/libfpv_amd_cpushim.so
build/asan/libfpv_amd_cpushim.so: tests/shim/fpvhip_oracle_shim.c oracle/fpv_oracle.c $(HOST_SRC) $(HOST_HDR)
	@mkdir -p build/asan
	gcc -std=c99 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Ioracle -I$(BROTLI_I) -c tests/shim/fpvhip_oracle_shim.c -o build/asan/shim.o
	gcc -std=c99 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -Ioracle -I$(BROTLI_I) -c oracle/fpv_oracle.c -o build/asan/oracle.o
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -Wl,-Bsymbolic $(HOST_SRC) build/asan/shim.o build/asan/oracle.o $(BROTLI) -o $@

# ThreadSanitizer run of the threaded host paths (native driver; CPU only).
tsan: build/tsan/driver build/tsan/columnar build/tsan/adapter_fail
	TSAN_OPTIONS=halt_on_error=1 build/tsan/driver
	TSAN_OPTIONS=halt_on_error=1 build/tsan/columnar
	TSAN_OPTIONS=halt_on_error=1 build/tsan/adapter_fail
build/tsan/adapter_fail: tests/adapter_fail_driver.cc build/tsan/driver
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=thread -Iinclude/columnar_batch tests/adapter_fail_driver.cc $(HOST_SRC) build/tsan/shim.o build/tsan/oracle.o $(BROTLI) -o $@
build/tsan/columnar: tests/columnar_driver.cc build/tsan/driver
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=thread -Iinclude/columnar_batch tests/columnar_driver.cc $(HOST_SRC) build/tsan/shim.o build/tsan/oracle.o $(BROTLI) -o $@
build/tsan/driver: tests/shim/tsan_driver.cc tests/shim/fpvhip_oracle_shim.c oracle/fpv_oracle.c $(HOST_SRC) $(HOST_HDR)
	@mkdir -p build/tsan
	gcc -std=c99 -O1 -g -fsanitize=thread -Iinclude -Ioracle -I$(BROTLI_I) -c tests/shim/fpvhip_oracle_shim.c -o build/tsan/shim.o
	gcc -std=c99 -O1 -g -fsanitize=thread -Ioracle -I$(BROTLI_I) -c oracle/fpv_oracle.c -o build/tsan/oracle.o
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=thread tests/shim/tsan_driver.cc $(HOST_SRC) build/tsan/shim.o build/tsan/oracle.o $(BROTLI) -o $@

clean:
	rm -rf $(LIBDIR) build tests/shim/*.so tests/shim/*.o tests/shim/*_cpushim
	$(MAKE) -C oracle clean

.PHONY: all product tools oracle shim refcli refcb columnar arrow asan tsan clean
