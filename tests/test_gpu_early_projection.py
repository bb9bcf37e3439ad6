"""GPU tests of the encoders' input projections enqueued ahead of the plan (ABI 6: stair_encoders_project, STAIR_PLAN_EXT_PROJECTION,
STAIR_RUN_PROJECTED; VideoNMN.early_projection).  The reference encodes clip and question before its interpreter looks at the program
(/root/reference/video_nmn/module_net.py:74-75); here the first half of both encoders runs before the host has packed the programs.
It is the same kernels on the same operands in another buffer: every result must be BIT-identical to the plan computing its own
projections -- logits, every parameter gradient, the weights after Adam -- for bf16 and fp32 features, shared clips, ragged clip
lengths, and through a captured plan's replay."""
import numpy as np
import pytest
import torch

from stair_amd import spec, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(config, seed=0):
    from stair_amd.module_net import VideoNMN
    m = VideoNMN(config)
    w = synth.make_weights(config, seed)
    m.load_state_dict({k: torch.from_numpy(w[k].copy()) for k in spec.state_dict_keys(config)})
    return m.to(DEV)


def _batch(config, n, bf16, T=64):
    qs = [synth.make_question(config, 3, i, T=T, forms=synth.ALL_FORMS) for i in range(n)]
    video = torch.stack([torch.as_tensor(q['video_features']) for q in qs]).to(DEV)
    if bf16:
        video = video.to(torch.bfloat16)
    question = torch.cat([torch.as_tensor(q['question']) for q in qs]).to(DEV)
    return ([q['nmn_program_list'] for q in qs], [q['prog_str_to_question_tokens'] for q in qs], video, question,
            [q['question'].shape[0] for q in qs], torch.tensor([q['answer'] for q in qs], dtype=torch.int32, device=DEV))


@pytest.mark.parametrize('bf16', [True, False])
@pytest.mark.parametrize('mode', ['plain', 'shared_clips', 'ragged'])
def test_forward_is_bit_identical(bf16, mode):
    config = dict(spec.DEFAULT_CONFIG)
    m = _model(config)
    progs, spans, video, question, q_lens, _ = _batch(config, 48, bf16)
    kw = {}
    if mode == 'shared_clips':
        video = video[:16].contiguous()
        kw['video_index'] = [i % 16 for i in range(48)]
    elif mode == 'ragged':
        vlen = np.random.RandomState(0).randint(9, 65, size=48).astype(np.int32)
        mask = (torch.arange(64, device=DEV)[None, :] < torch.as_tensor(vlen, device=DEV)[:, None]).unsqueeze(-1)
        video = (video * mask).contiguous()
        kw['video_len'] = vlen
    out = {}
    for early in (False, True):
        m.early_projection = early
        res = m.run_programs(progs, spans, video, question, q_lens, **kw)
        out[early] = (res.logits.clone(), res.pred.clone(), res.token_feature.clone(), res.info.workspace_bytes)
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
    assert torch.equal(out[False][2], out[True][2])
    assert out[True][3] < out[False][3]                   # the projection regions left the workspace


def test_training_step_is_bit_identical():
    from stair_amd.train import Trainer
    config = dict(spec.DEFAULT_CONFIG)
    progs, spans, video, question, q_lens, answers = _batch(config, 96, True)
    runs = {}
    for early in (False, True):
        m = _model(config)
        m.early_projection = early
        tr = Trainer(m, dropout=0.0)
        for _ in range(2):                                 # the second step runs on updated weights: the projection must read the CURRENT ones
            ce, res = tr.step(progs, spans, video, question, q_lens, answers)
        torch.cuda.synchronize()
        runs[early] = (ce.clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}, tr.flat_p.clone())
    assert torch.equal(runs[False][0], runs[True][0])
    for n in runs[False][1]:
        assert torch.equal(runs[False][1][n], runs[True][1][n]), n
    assert torch.equal(runs[False][2], runs[True][2])


def test_captured_plan_replays_with_its_own_projection_buffer():
    config = dict(spec.DEFAULT_CONFIG)
    m = _model(config)
    progs, spans, video, question, q_lens, _ = _batch(config, 32, True)
    m.early_projection = True
    res = m.run_programs(progs, spans, video, question, q_lens)
    ref = res.logits.clone()
    cap = res.capture_graph()
    # another batch rewrites the model's shared buffers; the captured plan must not care
    p2, s2, v2, q2, l2, _ = _batch(config, 40, True)
    m.run_programs(p2, s2, v2, q2, l2)
    logits, _ = cap.replay()
    torch.cuda.synchronize()
    assert torch.equal(logits, ref)
    # new inputs in place: the replay recomputes the projections from them
    video.copy_(video.flip(0))
    logits, _ = cap.replay()
    torch.cuda.synchronize()
    m.early_projection = False
    again = m.run_programs(progs, spans, video, question, q_lens).logits
    assert torch.equal(logits, again)
