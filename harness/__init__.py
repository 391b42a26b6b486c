"""Bench / test HARNESS, not product: a mirror of ARTDECO's mapper host code (harness/mapper.py: the call order of
SceneModel.render / render_from_id / optimization_step and the optimiser classes, pinned bit for bit to the reference's own
source by tests/test_mapper_host_logic.py and tests/test_optimizer_host_logic.py) and the synthetic workloads that
bench.py and the GPU tests drive the natives with.  In a deployment ARTDECO's own files play this role; nothing under
artdeco_amd/ imports this package."""
