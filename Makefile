# Top-level conveniences.  The product is built by `python -m panovlm_amd.build` (hipcc, gfx950); nothing here is needed for it.
#
#   make repin REFERENCE=/path/to/PanoVLM [REFERENCE_BUILD=/path/to/PanoVLM/build] [JOBS=8]
#
# One command on a machine where PanoVLM's dependencies exist (Eigen 3.4, PCL 1.10, Ceres 2.0, OpenCV 3.4, Boost, glog): builds PanoVLM itself with
# its own CMakeLists (the reference's CMakeLists.txt:6-60), builds tools/dump_reference_vectors.cpp against it, runs the REAL reference functions on
# the inputs of every fixture under tests/golden/, compares (tools/refvec.py compare: the table of DESIGN.md section 4), rewrites the fixtures'
# expectations from the reference's outputs (tagged with the reference's commit) and runs the CPU suite against them.  `make -n repin REFERENCE=...`
# prints the commands without running them (tests/test_golden_cpu.py checks that it does).
PYTHON ?= python3
JOBS ?= 8
REFERENCE_BUILD ?= $(REFERENCE)/build
REPIN_BUILD ?= build/repin

.PHONY: repin repin-table lib
lib:
	$(PYTHON) -m panovlm_amd.build

repin-table:
	$(PYTHON) tools/refvec.py table

repin:
	@test -n "$(REFERENCE)" || { echo "usage: make repin REFERENCE=/path/to/PanoVLM [REFERENCE_BUILD=...]"; exit 2; }
	@test -f "$(REFERENCE)/base/CostFunction.h" || { echo "REFERENCE=$(REFERENCE) is not a PanoVLM checkout (base/CostFunction.h not found)"; exit 2; }
	cmake -S "$(REFERENCE)" -B "$(REFERENCE_BUILD)" -DCMAKE_BUILD_TYPE=Release
	cmake --build "$(REFERENCE_BUILD)" -j $(JOBS)
	cmake -S tools/repin -B "$(REPIN_BUILD)" -DPANOVLM_ROOT="$(REFERENCE)" -DPANOVLM_BUILD="$(REFERENCE_BUILD)"
	cmake --build "$(REPIN_BUILD)" --target repin_regenerate -j $(JOBS)
	$(PYTHON) tools/refvec.py table
	@echo "std::sort order of equal keys: re-pinned against `$${CXX:-g++} --version | head -1` (tests/test_stdsort_cpu.py, compiled with that compiler)"
	$(PYTHON) -m pytest tests/test_stdsort_cpu.py -q
	$(PYTHON) -m pytest tests -q -m "not gpu"
	@echo "re-pinned: commit tests/golden/*.npz and the table above (profiles/), replace 'parity unpinned' in DESIGN.md section 4 and oracle/*.hpp"
	@echo "with a GPU: $(REPIN_BUILD)/ceres_adapter_check ceresadapter <scans.bin> 0.05 1.0   (integration/pvlm_ceres.hpp under the real Ceres)"
