"""Second opinion on the trunk oracle.  The reference takes its ResNet from torchvision 0.13.1, which is neither vendored nor
installed (oracle/resnet.py restates it: "parity unpinned").  `transformers` IS installed here and on the GPU box, and its
`ResNetModel` is an independent implementation of the same published architecture (v1.5: the stride sits on the 3x3 convolution
of a bottleneck, projection shortcut = 1x1 convolution + BN): the oracle's weights are copied into it key by key and the five
feature maps of models/resnet_encoder.py:115-126 must agree, in eval mode (running statistics) and in train mode (batch
statistics + the running-statistic updates).  This does not lift the "unpinned" label -- the reference's own dependency has
not run -- but a transcription error in the oracle's block arithmetic would have to be shared by both implementations."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf_model(num_layers, in_ch):
    from transformers import ResNetConfig, ResNetModel
    from oracle import resnet as R
    kind, depths = R.LAYERS[num_layers]
    exp = 1 if kind == "basic" else 4
    cfg = ResNetConfig(num_channels=in_ch, embedding_size=64, hidden_sizes=[64 * exp, 128 * exp, 256 * exp, 512 * exp], depths=depths,
                       layer_type=kind, hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    return ResNetModel(cfg)


def _copy(sd, model, num_layers):
    """oracle (torchvision) keys -> transformers keys"""
    from oracle import resnet as R
    tgt = model.state_dict()

    def put(dst, src):
        assert tgt[dst].shape == sd[src].shape, (dst, src, tuple(tgt[dst].shape), tuple(sd[src].shape))
        tgt[dst] = sd[src].clone()

    def put_bn(dst, src):
        for k in ("weight", "bias", "running_mean", "running_var"):
            put(f"{dst}.{k}", f"{src}.{k}")

    put("embedder.embedder.convolution.weight", "conv1.weight")
    put_bn("embedder.embedder.normalization", "bn1")
    for prefix, kind, _inpl, _planes, _stride, ds in R.block_plan(num_layers):
        li, bi = prefix.split(".")
        base = f"encoder.stages.{int(li[-1]) - 1}.layers.{bi}"
        for j in range(2 if kind == "basic" else 3):
            put(f"{base}.layer.{j}.convolution.weight", f"{prefix}.conv{j + 1}.weight")
            put_bn(f"{base}.layer.{j}.normalization", f"{prefix}.bn{j + 1}")
        if ds:
            put(f"{base}.shortcut.convolution.weight", f"{prefix}.downsample.0.weight")
            put_bn(f"{base}.shortcut.normalization", f"{prefix}.downsample.1")
    missing = model.load_state_dict(tgt, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


def _randomise_bn(sd, gen):
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
        elif ".bn" in k or k.startswith("bn") or "downsample.1" in k:
            if k.endswith("weight"):
                sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
            elif k.endswith("bias"):
                sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.1


@pytest.mark.parametrize("num_layers,in_ch", [(18, 3), (18, 6), (34, 3), (50, 3), (101, 3)])
@pytest.mark.parametrize("training", [False, True])
def test_oracle_trunk_matches_an_independent_resnet(num_layers, in_ch, training):
    from oracle import resnet as R
    gen = torch.Generator().manual_seed(num_layers * 10 + in_ch)
    sd = R.make_state_dict(num_layers, in_ch, seed=num_layers, with_fc=False)
    _randomise_bn(sd, gen)
    model = _hf_model(num_layers, in_ch).double()
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _copy(sd, model, num_layers)
    model.train(training)
    x = torch.randn(2, in_ch, 64, 96, generator=gen).double()
    with torch.no_grad():
        # transformers: hidden_states[0] = the embedder's output AFTER its max-pool, [1..4] = the stages; the oracle's feat0 is
        # relu(bn1(conv1 x)) BEFORE the pool -> recompute it from the embedder's convolution layer
        emb = model.embedder.embedder
        f0 = emb.activation(emb.normalization(emb.convolution(x)))
        if training:                                   # (the line above already updated bn1's running statistics once: undo)
            _copy(sd, model, num_layers)
        out = model(x, output_hidden_states=True)
        want = [f0] + list(out.hidden_states[1:])
    sd_run = {k: v.clone() for k, v in sd.items()}
    got = R.features(sd_run, x, num_layers, training=training)
    assert len(got) == 5
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape, (i, a.shape, b.shape)
        assert float((a - b).abs().max()) <= 1e-10 * max(1.0, float(b.abs().max())), f"feature {i}"
    if training:                                       # running statistics after one train-mode forward
        hf = model.state_dict()
        assert torch.allclose(hf["embedder.embedder.normalization.running_mean"], sd_run["bn1.running_mean"], atol=1e-12)
        last = R.block_plan(num_layers)[-1][0]
        li, bi = last.split(".")
        k = f"encoder.stages.{int(li[-1]) - 1}.layers.{bi}.layer.1.normalization.running_var"
        assert torch.allclose(hf[k], sd_run[f"{last}.bn2.running_var"], atol=1e-12)
