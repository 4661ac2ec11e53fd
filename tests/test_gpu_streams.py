"""Whole-stream parity on the GPU: product driver + HIP model vs what the reference driver + reference model produced."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from helpers import hip_model, stream_cases, run_stream_case
from mmduet_amd.inference import LiveInferForBenchmark

META = stream_cases()


@pytest.fixture(scope='module')
def model_f32():
    return hip_model('A', torch.float32)[0]


@pytest.mark.parametrize('name', list(META['cases']))
@pytest.mark.parametrize('k', [1, 4])
def test_stream_fp32_matches_reference(model_f32, name, k):
    case = META['cases'][name]
    d = run_stream_case(LiveInferForBenchmark, model_f32, name, case, META, frames_per_forward=k)
    assert len(d.debug_data_list) == case['T']
    for got, exp in zip(d.debug_data_list, case['debug_data']):
        assert got['time'] == pytest.approx(exp['time'])
        assert got['informative_score'] == pytest.approx(exp['informative_score'], abs=2e-4)
        assert got['relevance_score'] == pytest.approx(exp['relevance_score'], abs=2e-4)
    assert d.response_token_ids == case['generated']
    assert [(r['role'], r['time'], r['content']) for r in d.responses] == [(r['role'], pytest.approx(r['time']), r['content']) for r in case['responses']]
    assert len(d.past_key_values) == case['final_kv_len']
    assert [int(x) for x in d.generated_token_ids] == case['penalty_ids']
    if k > 1 and case['n_responses'] == 0:
        assert d.forward_calls <= -(-case['T'] // k) + len(case['conversation']) + 1


def test_stream_bf16_runs_and_stays_close():
    m = hip_model('A', torch.bfloat16)[0]
    case = META['cases']['grounding_q0']
    d = run_stream_case(LiveInferForBenchmark, m, 'grounding_q0', case, META, dtype=torch.bfloat16)
    for got, exp in zip(d.debug_data_list, case['debug_data']):
        assert got['informative_score'] == pytest.approx(exp['informative_score'], abs=5e-2)
        assert got['relevance_score'] == pytest.approx(exp['relevance_score'], abs=5e-2)
    assert len(d.past_key_values) == case['final_kv_len']
