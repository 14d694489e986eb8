"""Model definitions (nnet/models_zoo.py): audio-only, visual-only and audio-visual Efficient Conformer InterCTC models."""
from . import losses as L
from . import networks, optimizers, schedulers
from .model import Model


def _default_adam(model):
    """Adam(b=(0.9,0.98), eps=1e-9, wd=1e-6) + Noam(warmup 10k, dim 360, factor 2)  (nnet/models_zoo.py:172-174)"""
    lr = schedulers.NoamDecayScheduler(warmup_steps=10000, dim_decay=360, val_factor=2)
    return optimizers.Adam(params=model.parameters(), lr=lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)


class VisualEfficientConformerCE(Model):
    """nnet/models_zoo.py:33-62: the LRW word classifier -- visual encoder without InterCTC, head to `vocab_size` classes, logits averaged over time."""

    def __init__(self, vocab_size=500):
        super().__init__(name="Visual Efficient Conformer CE")
        self.encoder = networks.VisualEfficientConformerEncoder(vocab_size=vocab_size, interctc_blocks=[])

    def forward(self, inputs):
        from .. import ops
        x = self.encoder(inputs, lengths=None)[0]
        return ops.TimeMeanFn.apply(x) if x.is_cuda else x.mean(dim=1)

    def compile(self, losses=None, loss_weights=None, optimizer="Adam", metrics="default", decoders=None):
        from . import metrics as M
        if losses is None:
            losses = L.SoftmaxCrossEntropy()
        if metrics == "default":
            metrics = M.CategoricalAccuracy()
        if optimizer == "Adam":
            optimizer = _default_adam(self)
        super().compile(losses=losses, loss_weights=loss_weights, optimizer=optimizer, metrics=metrics, decoders=decoders)


class _InterCTCModel(Model):
    default_loss_weights = None

    def compile(self, losses=None, loss_weights="default", optimizer="Adam", metrics=None, decoders=None):
        if losses is None:
            losses = L.CTCLoss()
        if loss_weights == "default":
            lw = self.default_loss_weights
            loss_weights = dict(lw) if isinstance(lw, dict) else list(lw)
        if optimizer == "Adam":
            optimizer = _default_adam(self)
        super().compile(losses=losses, loss_weights=loss_weights, optimizer=optimizer, metrics=metrics, decoders=decoders)


class AudioEfficientConformerInterCTC(_InterCTCModel):
    """nnet/models_zoo.py:64-97"""
    default_loss_weights = [0.5 / 4, 0.5 / 4, 0.5 / 4, 0.5 / 4, 0.5]

    def __init__(self, vocab_size=256, att_type="patch", interctc_blocks=[3, 6, 10, 13]):
        super().__init__(name="Audio Efficient Conformer Inter CTC")
        self.encoder = networks.AudioEfficientConformerEncoder(vocab_size=vocab_size, att_type=att_type, interctc_blocks=interctc_blocks)

    def forward(self, inputs):
        x, lengths = inputs
        x, lengths, inter = self.encoder(x, lengths)
        out = {"outputs": [x, lengths]}
        out.update(inter)
        return out


class VisualEfficientConformerInterCTC(_InterCTCModel):
    """nnet/models_zoo.py:99-147 (test-time augmentation is an evaluation nicety outside the hot path)"""
    default_loss_weights = [0.5 / 3, 0.5 / 3, 0.5 / 3, 0.5]

    def __init__(self, vocab_size=256, interctc_blocks=[3, 6, 9], test_augments=None):
        super().__init__(name="Visual Efficient Conformer Inter CTC")
        assert test_augments is None
        self.encoder = networks.VisualEfficientConformerEncoder(vocab_size=vocab_size, interctc_blocks=interctc_blocks)

    def forward(self, inputs):
        video, video_lengths = inputs
        x, lengths, inter = self.encoder(video.permute(0, 4, 1, 2, 3), video_lengths)
        out = {"outputs": [x, lengths]}
        out.update(inter)
        return out


class AudioVisualEfficientConformerInterCTC(_InterCTCModel):
    """nnet/models_zoo.py:149-182"""
    default_loss_weights = {"v_ctc_2": 0.5 / 3, "v_ctc_5": 0.5 / 3, "a_ctc_7": 0.5 / 3, "a_ctc_10": 0.5 / 3, "f_ctc_1": 0.5 / 3, "outputs": 0.5}

    def __init__(self, vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2]):
        super().__init__(name="Audio-Visual Efficient Conformer Inter CTC")
        self.encoder = networks.AudioVisualEfficientConformerEncoder(vocab_size=vocab_size, v_interctc_blocks=v_interctc_blocks,
                                                                     a_interctc_blocks=a_interctc_blocks, f_interctc_blocks=f_interctc_blocks)

    def forward(self, inputs):
        video, video_len, audio, audio_len = inputs
        x, lengths, inter = self.encoder(video.permute(0, 4, 1, 2, 3), video_len, audio, audio_len)
        out = {"outputs": [x, lengths]}
        out.update(inter)
        return out
