# Top-level convenience targets; the real recipes live next to the sources.
all:
	$(MAKE) -C abft_sparse_cg_amd/csrc
	$(MAKE) -C abft_sparse_cg_amd/host
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -x -q -m "not gpu"

test-gpu: all
	python -m pytest tests -x -q -m gpu

bench: all
	python bench.py

clean:
	$(MAKE) -C abft_sparse_cg_amd/csrc clean
	$(MAKE) -C abft_sparse_cg_amd/host clean
	$(MAKE) -C oracle clean

.PHONY: all test-cpu test-gpu bench clean
