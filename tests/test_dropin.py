"""CPU: the drop-in boundary.  With this repository ahead of the reference
checkout on sys.path, the reference's UNCHANGED entry points must import, build their Trainer
(model through the HIP-path registry, criterion / optimizer / scheduler / data pipeline from
the reference's own files) and be launchable one process per GPU.  Each case runs in a fresh
interpreter: `import segmentron` rebinds process-wide module names."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "segmentron", "solver")),
                               reason="reference checkout not present (GPU box)")


def _run(code, env_extra=None, args=()):
    env = dict(os.environ)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import launch  # the environment tools/launch.py gives every rank (PYTHONPATH incl. shims)
    env = launch.build_env(0, 1, 29500, base={k: v for k, v in os.environ.items()
                                              if k != "PYTHONPATH"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", textwrap.dedent(code)] + list(args), env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return p.stdout


def test_without_reference_only_the_hot_path_resolves():
    out = _run("""
        import sys
        sys.path = [p for p in sys.path if 'reference' not in p]
        sys.argv = ['x']
        import segmentron, segmentron_amd
        assert segmentron.__reference_root__ is None
        from segmentron.config import cfg
        from segmentron.models.model_zoo import get_segmentation_model, MODEL_REGISTRY
        from segmentron.models.backbones import BACKBONE_REGISTRY
        from segmentron.data.dataloader import datasets
        assert cfg is segmentron_amd.config.cfg
        assert get_segmentation_model is segmentron_amd.get_segmentation_model
        assert set(MODEL_REGISTRY.get_list()) >= {'DeepLabV3_Plus', 'PSPNet', 'FCN', 'HRNet'}
        try:
            import segmentron.solver.loss
        except ModuleNotFoundError as e:
            assert 'SEGMENTRON_REFERENCE_ROOT' in str(e)
            print('OK')
    """, env_extra={"SEGMENTRON_REFERENCE_ROOT": ""})
    assert "OK" in out


@needs_ref
def test_reference_train_py_import_block_and_trainer_build(tmp_path):
    """Executes /root/reference/tools/train.py itself (its module body = the import block
    :17-29 and the Trainer class), then builds Trainer exactly as its __main__ block does, on a
    six-image synthetic Cityscapes tree; one batch is drawn through the reference's dataset +
    the torchvision stand-in."""
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(0)
    for split, n in (("train", 4), ("val", 2)):
        for i in range(n):
            d_img = tmp_path / "datasets" / "cityscapes" / "leftImg8bit" / split / "aachen"
            d_gt = tmp_path / "datasets" / "cityscapes" / "gtFine" / split / "aachen"
            d_img.mkdir(parents=True, exist_ok=True)
            d_gt.mkdir(parents=True, exist_ok=True)
            Image.fromarray(rng.randint(0, 255, (96, 192, 3), dtype=np.uint8)).save(
                d_img / ("aachen_%06d_000019_leftImg8bit.png" % i))
            Image.fromarray(rng.randint(0, 34, (96, 192), dtype=np.uint8)).save(
                d_gt / ("aachen_%06d_000019_gtFine_labelIds.png" % i))
    out = _run("""
        import importlib.util, os, sys, types
        ref, root = sys.argv[1], sys.argv[2]
        script = os.path.join(ref, 'tools', 'train.py')
        sys.argv = [script]
        spec = importlib.util.spec_from_file_location('ref_tools_train', script)
        train = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(train)            # sys.path.append(root_path) + the import block
        import segmentron, segmentron_amd, torch
        assert segmentron.__reference_root__ == ref
        assert sys.modules['segmentron.solver.loss'].__file__.startswith(ref)
        assert sys.modules['segmentron.utils.distributed'].__file__.startswith(ref)
        assert sys.modules['segmentron.data.dataloader'].__file__.startswith(ref)
        assert train.get_segmentation_model is segmentron_amd.get_segmentation_model
        assert train.cfg is segmentron_amd.config.cfg
        cfg = train.cfg
        cfg.update_from_file(os.path.join(ref, 'configs', 'cityscapes_deeplabv3_plus.yaml'))
        cfg.update_from_list(['TRAIN.BATCH_SIZE', '2', 'TRAIN.CROP_SIZE', '64', 'TRAIN.BASE_SIZE',
                              '96', 'DATASET.WORKERS', '0', 'TRAIN.BACKBONE_PRETRAINED', 'False',
                              'TRAIN.LOG_SAVE_DIR', os.path.join(root, 'log'), 'TRAIN.EPOCHS', '1'])
        cfg.PHASE = 'train'
        cfg.ROOT_PATH = root
        cfg.check_and_freeze()
        args = types.SimpleNamespace(no_cuda=True, local_rank=0, resume=None, log_iter=10,
                                     val_epoch=1, skip_val=True, config_file='x', opts=[])
        train.default_setup(args)
        assert args.device == 'cpu' and args.distributed is False
        trainer = train.Trainer(args)
        m = trainer.model
        assert type(m).__module__ == 'segmentron_amd.models.deeplabv3_plus'
        assert type(trainer.criterion).__name__ == 'MixSoftmaxCrossEntropyLoss'
        # optimizer.py:16-30 walked model.encoder / model.decoder and set the encoder BN eps
        assert m.encoder.block4.sep_conv1.block.bn_depth.eps == cfg.MODEL.BN_EPS_FOR_ENCODER
        assert len(trainer.optimizer.param_groups) == 2
        assert trainer.optimizer.param_groups[1]['lr'] == cfg.SOLVER.LR * 10
        images, targets, names = next(iter(trainer.train_loader))
        assert images.shape == (2, 3, 64, 64) and images.dtype == torch.float32
        assert targets.shape == (2, 64, 64) and int(targets.min()) >= -1 and int(targets.max()) < 19
        # the forward needs the HIP device: on CPU it must fail loudly, not fall back
        try:
            m(images)
            raise SystemExit('CPU forward did not raise')
        except RuntimeError as e:
            assert 'HIP' in str(e)
        print('TRAINER_OK', trainer.max_iters)
    """, args=[REF, str(tmp_path)])
    assert "TRAINER_OK 2" in out


@needs_ref
def test_reference_eval_and_demo_import_blocks():
    out = _run("""
        import ast, os, sys
        ref = sys.argv[1]
        sys.path.append(ref)
        for name in ('eval.py', 'demo.py'):
            src = open(os.path.join(ref, 'tools', name)).read()
            tree = ast.parse(src)
            imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
            mod = ast.Module(body=imports, type_ignores=[])
            exec(compile(mod, name, 'exec'), {'__name__': 'probe'})
        import segmentron_amd
        from segmentron.models.model_zoo import get_segmentation_model
        assert get_segmentation_model is segmentron_amd.get_segmentation_model
        print('OK')
    """, args=[REF])
    assert "OK" in out


def test_launcher_sets_rank_env_and_local_rank_arg(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(textwrap.dedent("""
        import argparse, os, sys
        ap = argparse.ArgumentParser()
        ap.add_argument('--config-file')
        ap.add_argument('--local_rank', type=int, default=0)
        ap.add_argument('opts', nargs=argparse.REMAINDER)
        a = ap.parse_args()
        assert a.local_rank == int(os.environ['LOCAL_RANK']) == int(os.environ['RANK'])
        assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'
        assert a.opts == ['TRAIN.BATCH_SIZE', '2'] and a.config_file == 'c.yaml'
        import segmentron_amd  # this repository is ahead on PYTHONPATH
        import torch.distributed as dist
        dist.init_process_group('gloo', init_method='env://')
        import torch
        t = torch.ones(1) * (dist.get_rank() + 1)
        dist.all_reduce(t)
        assert t.item() == 3.0
        dist.destroy_process_group()
        open(os.path.join(os.path.dirname(__file__), 'rank%d.ok' % a.local_rank), 'w').write('ok')
    """))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch.py"), "--nproc", "2",
                        str(script), "--config-file", "c.yaml", "TRAIN.BATCH_SIZE", "2"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
