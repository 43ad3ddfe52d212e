# Builds the gfx950 C-ABI library in-tree (the .so travels to the GPU box, it is git-ignored).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := speakerguard_amd/csrc
OBJ   := build/obj
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS))
LIB   := speakerguard_amd/libspeakerguard_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(EXTRA)

ORACLE_SO := oracle/libconv_chain.so

# the checker's C part is built along (test infrastructure), but a host without gcc / OpenMP / -mfma must not fail the
# PRODUCT build because of it: oracle/conv_chain.py builds it lazily on first use anyway
all: $(LIB)
	-@$(MAKE) --no-print-directory oracle

# the CPU checker's C part (test infrastructure: never linked into the product library)
$(ORACLE_SO): oracle/conv_chain.c
	gcc -O2 -mfma -fopenmp -ffp-contract=off -shared -fPIC $< -o $@ -lm

oracle: $(ORACLE_SO)

$(OBJ)/%.o: $(CSRC)/%.hip $(CSRC)/sg_internal.h $(CSRC)/fft512.h $(CSRC)/fft512t.h $(CSRC)/loss_device.h include/speakerguard_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# ---- CPU sanitizer build of the C-ABI's HOST half (runs in the GPU-less container) ------------
# hipcc --cuda-host-only: no device code is generated or needed; the HIP runtime is replaced by a host-memory double
# (tests/native/hip_host_double.cpp: test infrastructure) so that argument validation, table builders, weight folding /
# packing, workspace sizing and launch selection all execute under AddressSanitizer + UBSan.  No GPU sanitizer involved.
ASAN_DIR   := build/asan
ASAN_FLAGS := --cuda-host-only -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-sanitize-recover=all \
              -fno-omit-frame-pointer -Iinclude -Wall -Wno-unused-function
ASAN_OBJS  := $(patsubst $(CSRC)/%.hip,$(ASAN_DIR)/%.o,$(SRCS))
ASAN_EXE   := $(ASAN_DIR)/abi_asan_driver

$(ASAN_DIR)/%.o: $(CSRC)/%.hip $(CSRC)/sg_internal.h $(CSRC)/fft512.h $(CSRC)/fft512t.h $(CSRC)/loss_device.h include/speakerguard_hip.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@

$(ASAN_DIR)/hip_host_double.o: tests/native/hip_host_double.cpp
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -x hip -c $< -o $@

$(ASAN_DIR)/abi_asan_driver.o: tests/native/abi_asan_driver.cpp include/speakerguard_hip.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -x c++ -c $< -o $@

# (the host objects reference their embedded-code-object symbols even when none is embedded: define them empty)
$(ASAN_EXE): $(ASAN_OBJS) $(ASAN_DIR)/hip_host_double.o $(ASAN_DIR)/abi_asan_driver.o
	nm -u $(ASAN_OBJS) | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u | sed 's/.*/char &[8];/' > $(ASAN_DIR)/fatbin_syms.c
	gcc -c $(ASAN_DIR)/fatbin_syms.c -o $(ASAN_DIR)/fatbin_syms.o
	/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address,undefined -o $@ $^ $(ASAN_DIR)/fatbin_syms.o

asan: $(ASAN_EXE)
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 $(ASAN_EXE)

clean:
	rm -rf build $(LIB) $(ORACLE_SO)

.PHONY: all clean oracle asan
