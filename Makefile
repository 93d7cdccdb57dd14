# Convenience wrapper: `make` at the repository root builds the HIP library and the CPU oracle
# (the same two steps as __graft_entry__.build()).
all:
	$(MAKE) -C ilqr_iterative_tasks_amd/csrc
	$(MAKE) -C oracle

clean:
	$(MAKE) -C ilqr_iterative_tasks_amd/csrc clean
	$(MAKE) -C oracle clean
