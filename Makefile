# Plain-make entry points for a maintainer who integrates libmi355zk.so without the Python tooling.
#   make lib        libmi355zk.so for gfx950 (the same command scroll-prover_amd/build.py runs; ~2 min)
#   make oracle     the CPU oracle (test infrastructure only)
#   make test-cpu   the GPU-less test suite          make test-gpu   the -m gpu suite (needs an MI355X)
#   make bench      the headline measurement (one JSON line)
HIPCC ?= /opt/rocm/bin/hipcc
CSRC := scroll-prover_amd/csrc
LIB := scroll-prover_amd/libmi355zk.so
DEPS := $(wildcard $(CSRC)/*.hip $(CSRC)/*.cuh $(CSRC)/*.inc) include/mi355zk.h

.PHONY: lib oracle test-cpu test-gpu bench clean
lib: $(LIB)
$(LIB): $(DEPS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DNDEBUG -Wno-unused-result -o $@ $(CSRC)/capi.hip
oracle:
	$(MAKE) -C oracle
test-cpu: lib oracle
	python -m pytest tests -q -m "not gpu"
test-gpu: lib oracle
	python -m pytest tests -q -m gpu
bench: lib oracle
	python bench.py
clean:
	rm -f $(LIB); $(MAKE) -C oracle clean
