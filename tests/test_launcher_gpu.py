"""train_mask_bev_amd.py (the counterpart of /root/reference: train_mask_bev.py:34-119) runs the built-in loop on an
MI355X: YAML -> MaskBevModule.from_config -> arena + HIP-graph step -> checkpoints named like the reference's
ModelCheckpoint -> --test reloads the best one."""
import re

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_launcher_trains_two_steps_on_synthetic_data(tmp_path, capsys, dtype):
    import train_mask_bev_amd as launcher
    from mask_bev_amd import synthetic
    kw = dict(synthetic.module_kwargs('smoke_96', 2, compute_dtype=dtype), dataset='synthetic',
              synthetic_points=6000, limit_val_batches=0.0, x_range=[-12, 12], y_range=[-12, 12], z_range=[-3, 1])
    cfg = tmp_path / 'smoke_96.yml'
    cfg.write_text(yaml.safe_dump(kw))
    ck = tmp_path / 'ckpt'
    rc = launcher.main(['--config', str(cfg), '--train', '--synthetic', '--max-epochs', '2', '--steps-per-epoch', '2',
                        '--checkpoint-root', str(ck)])
    assert rc == 0
    out = capsys.readouterr().out
    losses = [float(x) for x in re.findall(r'train_loss ([0-9.]+)', out)]
    assert len(losses) == 2 and all(torch.isfinite(torch.tensor(losses)))
    files = sorted(p.name for p in (ck / 'smoke_96').iterdir())
    assert 'last.ckpt' in files and any(re.match(r'smoke_96-epoch=\d\d-train_loss=[0-9.]+\.ckpt', f) for f in files)
    sd = torch.load(ck / 'smoke_96' / 'last.ckpt', weights_only=False)
    assert set(sd) >= {'state_dict', 'hyper_parameters', 'optimizer_states', 'epoch'}
    if dtype == 'fp16':                    # the loss-scaler state travels with the optimizer state
        assert sd['optimizer_states'][0]['loss_scaler']['scale'] >= 1.0
    # --test picks the best checkpoint by the loss in its file name (train_mask_bev.py:57-64) and reloads it
    rc = launcher.main(['--config', str(cfg), '--test', '--synthetic', '--checkpoint-root', str(ck)])
    assert rc == 0 and 'Testing from' in capsys.readouterr().out


def test_epoch_train_loss_is_the_mean_of_the_step_losses_in_graph_mode(tmp_path, capsys, monkeypatch):
    """The graphed step returns one static tensor that every replay overwrites; the launcher's epoch `train_loss`
    (what ReduceLROnPlateau, ModelCheckpoint and its file name monitor, mask_bev_module.py:161-166,296) must still be
    the mean over the epoch's steps, not the last step's value."""
    import train_mask_bev_amd as launcher
    from mask_bev_amd import graph, synthetic
    seen = []
    real_step = graph.GraphedTrainStep.step

    def recording_step(self, batch):
        loss = real_step(self, batch)
        seen.append(float(loss))           # read before the next replay overwrites the static tensor
        return loss

    monkeypatch.setattr(graph.GraphedTrainStep, 'step', recording_step)
    kw = dict(synthetic.module_kwargs('smoke_96', 2, compute_dtype='bf16'), dataset='synthetic',
              synthetic_points=6000, limit_val_batches=0.0, x_range=[-12, 12], y_range=[-12, 12], z_range=[-3, 1])
    cfg = tmp_path / 'smoke_96.yml'
    cfg.write_text(yaml.safe_dump(kw))
    rc = launcher.main(['--config', str(cfg), '--train', '--synthetic', '--max-epochs', '2', '--steps-per-epoch', '3',
                        '--checkpoint-root', str(tmp_path / 'ckpt')])
    assert rc == 0
    epoch_losses = [float(x) for x in re.findall(r'train_loss ([0-9.]+)', capsys.readouterr().out)]
    assert len(seen) == 6 and len(epoch_losses) == 2
    for e in range(2):
        steps = seen[3 * e:3 * e + 3]
        assert max(steps) - min(steps) > 1e-4 * max(steps)          # the steps do differ (different batches)
        assert epoch_losses[e] == pytest.approx(sum(steps) / 3, rel=2e-5)
