# Convenience targets; the driver-facing entry points are __graft_entry__.build()/smoke() and bench.py.
all:
	python -c "import __graft_entry__ as g; g.build()"

# the whole CPU suite, the five-minute scatter known answer included.  The log goes to an
# untracked file and the recipe fails when pytest does; `make pin-kats` makes the run the
# tracked record oracle/pins/full_kats.log.
SHELL := /bin/bash
test-cpu: all
	set -o pipefail; NEUTRAL_FULL_KATS=1 python -m pytest tests -x -q -rA -m "not gpu" 2>&1 | tee oracle/pins/_latest.log

# (the tracked record is made from a clean tree only: what it names is what was tested)
pin-kats:
	@git diff --quiet && git diff --cached --quiet || { echo "pin-kats: the tree has uncommitted changes: commit first"; exit 1; }
	$(MAKE) test-cpu
	{ echo "# make test-cpu at $$(git rev-parse --short HEAD), $$(date -u +%Y-%m-%dT%H:%MZ)"; \
	  echo "# oracle/neutral_oracle.c sha256 $$(sha256sum oracle/neutral_oracle.c | cut -c1-16)"; \
	  grep -E "PASSED|FAILED|SKIPPED|passed|failed|tally=" oracle/pins/_latest.log; } > oracle/pins/full_kats.log

test-gpu: all
	python -m pytest tests -x -q -m gpu

bench: all
	python bench.py

clean:
	$(MAKE) -C neutral_amd clean
	$(MAKE) -C oracle clean
	rm -rf integration/_dropin

.PHONY: all test-cpu pin-kats test-gpu bench clean
