"""CPU parity gate: the oracle, driven through the product's host-side file contract
(execute_demultiplexing with the oracle injected as classifier), must reproduce the
reference's byte-exact golden outputs (test/results/{demo1_R1,demo1_R2,demo2}) and the
expectations of its small integration tests.  This is what pins the oracle; the GPU tests
then hold the HIP path to the oracle."""
import functools

import pytest

import helpers as H

run = functools.partial(H.bdx.execute_demultiplexing, _classifier_factory=H.oracle_factory)


def test_demo1_R1_golden(tmp_path):
    assert H.scenario_demo1_R1(run, str(tmp_path)) == 24


def test_demo1_R2_golden(tmp_path):
    assert H.scenario_demo1_R2(run, str(tmp_path)) == 24


def test_demo2_golden(tmp_path):
    # the decisive one: weighted indel, min_delta, reverse-complement, 2306 mutated reads
    assert H.scenario_demo2(run, str(tmp_path)) == 76


@pytest.mark.parametrize("algorithm", ["exact", "hamming"])
def test_demo1_R1_other_modes(tmp_path, algorithm):
    # BASELINE config 1 ("tiny set, :exact, CPU path"): unmutated cores -> same goldens
    assert H.scenario_demo1_modes(run, str(tmp_path), algorithm) == 24


@pytest.mark.parametrize("scenario", H.SCENARIOS_SMALL, ids=[s.__name__ for s in H.SCENARIOS_SMALL])
def test_reference_integration_scenarios(tmp_path, scenario):
    scenario(run, str(tmp_path))


def test_small_batches_preserve_order(tmp_path):
    # the reference proves chunk re-ordering with a 4-thread rerun (runtests.jl:30-35);
    # here: many tiny device batches must give the same files as one big batch
    r = functools.partial(run, _batch_reads=3)
    assert H.scenario_demo1_R1(r, str(tmp_path)) == 24
