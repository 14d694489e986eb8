"""A config in the style of the reference's configs/LRS23/AV/EffConfInterCTC.py (model + compile + datasets + training options), with the synthetic
LRS-shaped dataset of nnet.datasets (no corpus on disk) and small sizes: used by tests/test_entry_point.py through main.py."""
import os
import tempfile

import nnet
import torch

vocab_size = 256
loss_weights = {"v_ctc_2": 0.5 / 3, "v_ctc_5": 0.5 / 3, "a_ctc_7": 0.5 / 3, "a_ctc_10": 0.5 / 3, "f_ctc_1": 0.5 / 3, "outputs": 0.5}

batch_size = 4
accumulated_steps = 1
eval_training = False
precision = torch.bfloat16
epochs = 1
recompute_metrics = True          # evaluation: the word error rate of the whole set from the gathered hypotheses (nnet/model.py:899-931), not the mean of per-batch rates
callback_path = os.environ.get("AVEC_TEST_CALLBACKS", os.path.join(tempfile.gettempdir(), "avec_callbacks", "av_synthetic"))

model = nnet.AudioVisualEfficientConformerInterCTC(vocab_size=vocab_size, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2])
model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False),
              decoders={"outputs": nnet.CTCGreedySearchDecoder()}, metrics={"outputs": nnet.WordErrorRate()}, loss_weights=loss_weights)

collate_fn = nnet.CollateFn(inputs_params=[{"axis": 0, "padding": True}, {"axis": 3}, {"axis": 1, "padding": True}, {"axis": 4}],
                            targets_params=({"axis": 2, "padding": True}, {"axis": 5}))
training_dataset = nnet.datasets.LRS(batch_size=batch_size, collate_fn=collate_fn, version="LRS2", mode="pretrain+train+val", video_max_length=100,
                                     align=True, num_synthetic=12, seed=0)
evaluation_dataset = [nnet.datasets.LRS(batch_size=batch_size, collate_fn=collate_fn, version="LRS2", mode="test", num_synthetic=8, seed=1)]
