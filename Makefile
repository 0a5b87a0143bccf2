# Repo-level helper targets (the library itself is built by libfluid_amd/build.py / __graft_entry__.build()).
PYTHON ?= python3
ASAN_LIB := $(shell gcc -print-file-name=libasan.so)

# CPU-side sanitizer run: the plain-C oracle and the pure-host C++ of libfluid_amd/host/ (formats; the drivers that need a GPU are
# compiled with the same flags by the GPU tests when LFA_HOST_CXXFLAGS is set) under AddressSanitizer + UBSan, driven by the CPU
# test-suite. (GPU AddressSanitizer is not available on the pool: the HIP side is covered by its parity tests.)
asan:
	$(MAKE) -C oracle asan
	LFA_ORACLE_SO=$(CURDIR)/oracle/_asan/liboracle.so LD_PRELOAD=$(ASAN_LIB) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
	UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 LFA_HOST_CXXFLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined -g" \
	$(PYTHON) -m pytest tests -q -m "not gpu" -p no:cacheprovider

.PHONY: asan
