"""The multi-GPU test engine without a GPU (reference: detectron/core/test_engine_wsl.py:148-200,
detectron/utils/subprocess.py:40-107): range split, child command line and environment, the
parent's collation of the children's range files, and the detections.pkl schema a reference-side
reader (tools/reval.py, tools/visualize_results.py) expects.  The children here are a stand-in
`test_net_wsl.py` that writes deterministic boxes per GLOBAL image index, so "two ranges collated"
must equal "one run over all images"."""
import json
import os
import pickle
import sys
import textwrap

import numpy as np
import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, 'na-fwebsod_amd', 'configs', 'flickr_voc', 'na_wsddn_V-16-C5_1x.yaml')

FAKE_BINARY = textwrap.dedent('''
    import os, pickle, sys
    import numpy as np
    sys.path.insert(0, %r)
    from detectron.core.config import assert_and_infer_cfg, cfg, merge_cfg_from_file, merge_cfg_from_list, get_output_dir
    from detectron.core import test_engine_wsl as te
    a = sys.argv[1:]
    s, e = int(a[a.index('--range') + 1]), int(a[a.index('--range') + 2])
    cfg_file = a[a.index('--cfg') + 1]
    opts = a[a.index('--cfg') + 2:]
    merge_cfg_from_file(cfg_file)
    merge_cfg_from_list(opts)
    assert_and_infer_cfg()
    assert cfg.NUM_GPUS == 1 and os.environ['HIP_VISIBLE_DEVICES'] in ('0', '1', '5')
    assert 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ
    print('child range', s, e, 'gpu', os.environ['HIP_VISIBLE_DEVICES'], 'weights', cfg.TEST.WEIGHTS)
    nc = cfg.MODEL.NUM_CLASSES
    ab, asg, akp = te.empty_results(nc, e - s)
    for i in range(s, e):
        for j in range(1, nc):
            n = (i * 7 + j) %% 4
            ab[j][i - s] = (np.arange(n * 5, dtype=np.float32).reshape(n, 5) + 100 * i + j)
    name, _pf = te.get_inference_dataset(0, is_parent=False)
    te.save_detections(os.path.join(get_output_dir(name, training=False),
                                    'detection_range_%%s_%%s.pkl' %% (s, e)), ab, asg, akp)
''') % os.path.join(ROOT, 'na-fwebsod_amd')


def test_split_ranges_and_gpu_inds():
    from detectron.utils import subprocess as su
    assert su.split_ranges(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]      # np.array_split
    assert su.split_ranges(2, 4) == [(0, 1), (1, 2)]                        # empty shares dropped
    assert su.split_ranges(4952, 8)[0] == (0, 619) and su.split_ranges(4952, 8)[-1] == (4333, 4952)
    assert su.visible_gpu_inds({}, 4) == [3, 2, 1, 0]                       # reference: reversed(range(N))
    assert su.visible_gpu_inds({'HIP_VISIBLE_DEVICES': '2,5'}, 2) == [2, 5]
    assert su.visible_gpu_inds({'CUDA_VISIBLE_DEVICES': '0,0'}, 2) == [0, 0]
    with pytest.raises(AssertionError):
        su.visible_gpu_inds({'HIP_VISIBLE_DEVICES': '0,-1'}, 2)
    # UUID-style tokens pass through as strings; a short list starts fewer children
    # (with a warning); ROCR_VISIBLE_DEVICES is consulted when the HIP list is unset
    assert su.visible_gpu_inds({'HIP_VISIBLE_DEVICES': 'GPU-1a2b,3'}, 2) == ['GPU-1a2b', 3]
    assert su.visible_gpu_inds({'HIP_VISIBLE_DEVICES': '4'}, 8) == [4]
    assert su.visible_gpu_inds({'ROCR_VISIBLE_DEVICES': '2,3'}, 8) == [1, 0]
    assert su.child_env({'PATH': '/bin'}, 'GPU-1a2b')['HIP_VISIBLE_DEVICES'] == 'GPU-1a2b'


def test_child_command_and_environment(cfgmod):
    from detectron.utils import subprocess as su
    from detectron.datasets import dataset_catalog
    cmd = su.child_command('/x/tools/test_net_wsl.py', 3, 6, '/o/detection_range_config.yaml',
                           ['TEST.DATASETS', '("voc_2007_test",)', 'TEST.WEIGHTS', '/w.pkl'])
    assert cmd[0] == sys.executable
    assert cmd[1:] == ['/x/tools/test_net_wsl.py', '--range', '3', '6', '--cfg',
                       '/o/detection_range_config.yaml', 'NUM_GPUS', '1', 'TEST.DATASETS',
                       '("voc_2007_test",)', 'TEST.WEIGHTS', '/w.pkl']
    dataset_catalog.register('toy_reg', '/im', '/ann.json')
    env = su.child_env({'PATH': '/bin', 'RANK': '3', 'WORLD_SIZE': '8', 'LOCAL_RANK': '3',
                        'MASTER_ADDR': 'h', 'CUDA_VISIBLE_DEVICES': '0,1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'}, 5)
    assert env['HIP_VISIBLE_DEVICES'] == '5' and env['PATH'] == '/bin'
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'CUDA_VISIBLE_DEVICES'):
        assert k not in env
    assert json.loads(env['NAWS_DATASET_REGISTRY'])['toy_reg']['annotation_file'] == '/ann.json'


def _cfg(cfgmod, tmp_path, n_images, gpus):
    cfgmod.reset_cfg()
    cfgmod.merge_cfg_from_file(YAML)
    cfgmod.merge_cfg_from_list(['OUTPUT_DIR', str(tmp_path), 'NUM_GPUS', gpus, 'TEST.DATASETS', '()',
                                'TEST.PROPOSAL_FILES', '()', 'NAWS.SYNTHETIC_TEST_IMAGES', n_images,
                                'MODEL.NUM_CLASSES', 4])
    cfgmod.assert_and_infer_cfg()


def test_two_ranges_collate_to_the_single_run(cfgmod, tmp_path, monkeypatch):
    from detectron.core import test_engine_wsl as te
    import detectron.utils.env as envu
    fake_dir = tmp_path / 'tools'
    fake_dir.mkdir()
    (fake_dir / 'test_net_wsl.py').write_text(FAKE_BINARY)
    monkeypatch.setattr(envu, 'get_runtime_dir', lambda: str(fake_dir))
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1')
    monkeypatch.setenv('RANK', '0')               # the parent of a torchrun job: must not leak
    monkeypatch.setenv('WORLD_SIZE', '2')
    weights = tmp_path / 'model_final.pkl'       # (the fake children never open it)
    weights.write_bytes(b'')
    # one "child" over everything = the single run
    _cfg(cfgmod, tmp_path / 'single', 5, 1)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0')
    res1 = te.run_inference(str(weights), multi_gpu_testing=True)
    single = pickle.load(open(os.path.join(str(tmp_path / 'single'), 'test', 'synthetic',
                                           'generalized_wsl', 'detections.pkl'), 'rb'))
    # two children, ranges [0, 3) and [3, 5)
    _cfg(cfgmod, tmp_path / 'double', 5, 2)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1')
    res2 = te.run_inference(str(weights), multi_gpu_testing=True)
    out = os.path.join(str(tmp_path / 'double'), 'test', 'synthetic', 'generalized_wsl')
    assert sorted(f for f in os.listdir(out) if f.endswith('.pkl')) == \
        ['detection_range_0_3.pkl', 'detection_range_3_5.pkl', 'detections.pkl']
    assert os.path.exists(os.path.join(out, 'detection_range_config.yaml'))
    assert 'child range 3 5 gpu 1' in open(os.path.join(out, 'detection_range_3_5.stdout')).read()
    double = pickle.load(open(os.path.join(out, 'detections.pkl'), 'rb'))
    assert res1 == res2 and res2['synthetic']['box']['num_images'] == 5
    for key in ('all_boxes', 'all_segms', 'all_keyps'):
        assert len(double[key]) == 4 and len(double[key][0]) == 0        # class 0: the reference's []
        for j in range(1, 4):
            assert len(double[key][j]) == 5
            for i in range(5):
                assert np.array_equal(np.asarray(double[key][j][i]), np.asarray(single[key][j][i])), (key, j, i)
    assert np.array_equal(double['all_boxes'][2][4], np.arange(10, dtype=np.float32).reshape(2, 5) + 402)


def test_detections_pickle_is_the_reference_schema(cfgmod, tmp_path):
    """Keys the reference reader indexes (tools/reval.py:85-98: dets['cfg'] merged key by key into
    its own cfg, dets['all_boxes'], ['all_segms'], ['all_keyps']); `cfg` is a yaml mapping made of
    reference keys only - every path in it exists in the cfg tree captured from the imported
    reference (tests/golden/reference_cfgs.json) or in the reference's defaults - and round-trips."""
    from detectron.core import test_engine_wsl as te
    from detectron.core.config import cfg
    _cfg(cfgmod, tmp_path, 2, 1)
    ab, asg, akp = te.empty_results(4, 2)
    ab[1][0] = np.ones((3, 5), np.float32)
    f = str(tmp_path / 'detections.pkl')
    te.save_detections(f, ab, asg, akp)
    det = pickle.load(open(f, 'rb'))
    assert set(det) == {'all_boxes', 'all_segms', 'all_keyps', 'cfg'}
    assert isinstance(det['cfg'], str)
    # every mapping carries the reference's AttrDict tag (its unsafe loader rebuilds AttrDicts,
    # which merge_cfg_from_cfg insists on); this package's loader reads the same text
    assert det['cfg'].startswith('!!python/object/new:detectron.utils.collections.AttrDict')
    from detectron.core.config import load_cfg
    tree = load_cfg(det['cfg'])
    assert type(tree) is dict and type(tree['TEST']['BBOX_AUG']) is dict
    assert 'NAWS' not in tree and tree['MODEL']['NUM_CLASSES'] == 4
    assert tree['TEST']['BBOX_AUG']['ENABLED'] == cfg.TEST.BBOX_AUG.ENABLED
    gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_cfgs.json')))['flickr_coco']

    def paths(t, pre=()):
        for k, v in t.items():
            if isinstance(v, dict):
                for p in paths(v, pre + (k,)):
                    yield p
            else:
                yield pre + (k,)
    gold_paths = set(paths(gold))
    ours = set(paths(tree))
    # the golden capture holds the subtrees the WSL yamls touch; every key of ours inside those
    # subtrees must exist there
    tops = {p[0] for p in gold_paths}
    unknown = sorted(p for p in ours if p[0] in tops and p not in gold_paths)
    assert not unknown, unknown
    # class-major lists of per-image arrays / empty lists, as extend_results leaves them
    assert det['all_boxes'][1][0].shape == (3, 5) and det['all_boxes'][1][1] == []
    assert det['all_segms'][1] == [[], []] and det['all_keyps'][3] == [[], []]


def test_failed_child_stops_the_parent(cfgmod, tmp_path, monkeypatch):
    from detectron.core import test_engine_wsl as te
    import detectron.utils.env as envu
    fake_dir = tmp_path / 'tools'
    fake_dir.mkdir()
    (fake_dir / 'test_net_wsl.py').write_text('import sys\nprint("boom")\nsys.exit(3)\n')
    monkeypatch.setattr(envu, 'get_runtime_dir', lambda: str(fake_dir))
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1')
    _cfg(cfgmod, tmp_path, 4, 2)
    with pytest.raises(AssertionError, match='Range subprocess failed'):
        te.run_inference('', multi_gpu_testing=True)


def test_failed_later_child_terminates_its_siblings(cfgmod, tmp_path, monkeypatch):
    """child 1 fails while child 0 is still running - the parent notices without
    waiting for child 0 to finish, and no child is left behind."""
    import time
    from detectron.core import test_engine_wsl as te
    import detectron.utils.env as envu
    fake_dir = tmp_path / 'tools'
    fake_dir.mkdir()
    (fake_dir / 'test_net_wsl.py').write_text(textwrap.dedent('''
        import os, sys, time
        a = sys.argv
        start = int(a[a.index('--range') + 1])
        if start > 0:
            sys.exit(7)
        open(os.path.join(%r, 'child0.pid'), 'w').write(str(os.getpid()))
        for _ in range(600):
            print('still running', flush=True)
            time.sleep(0.1)
    ''') % str(tmp_path))
    monkeypatch.setattr(envu, 'get_runtime_dir', lambda: str(fake_dir))
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1')
    _cfg(cfgmod, tmp_path, 4, 2)
    t0 = time.time()
    with pytest.raises(AssertionError, match='Range subprocess failed'):
        te.run_inference('', multi_gpu_testing=True)
    assert time.time() - t0 < 30            # (child 0 alone would run for a minute)
    pid = int(open(str(tmp_path / 'child0.pid')).read())
    time.sleep(0.2)
    with pytest.raises(OSError):
        os.kill(pid, 0)                     # terminated and reaped


def test_missing_weights_file_is_an_error(cfgmod, tmp_path):
    """a non-empty TEST.WEIGHTS that does not exist must not fall through to
    the randomly initialised model - neither in the parent (before any child starts), nor in a
    child / single process, nor in the CLI after its --wait loop gave up."""
    from detectron.core import test_engine_wsl as te
    _cfg(cfgmod, tmp_path, 2, 1)
    missing = str(tmp_path / 'model_final.pkl')
    with pytest.raises(FileNotFoundError, match='TEST.WEIGHTS'):
        te.run_inference(missing)
    with pytest.raises(FileNotFoundError):
        te.run_inference(missing, multi_gpu_testing=True)
    with pytest.raises(FileNotFoundError):
        te.run_inference(missing, ind_range=(0, 1))
    with pytest.raises(FileNotFoundError):
        te.initialize_model_from_cfg(missing)
    te.check_weights_file('')               # the synthetic smoke path stays available
    # the CLI: no --wait -> immediately
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'na-fwebsod_amd', 'tools', 'test_net_wsl.py'),
                        '--cfg', YAML, '--wait', '', 'OUTPUT_DIR', str(tmp_path), 'TEST.WEIGHTS', missing],
                       capture_output=True, text=True)
    assert r.returncode != 0 and 'FileNotFoundError' in r.stderr, r.stderr[-1500:]
    assert not any(f.endswith('.pkl') for _d, _s, fs in os.walk(str(tmp_path)) for f in fs)


@pytest.mark.skipif(not os.path.isdir('/root/reference/detectron'),
                    reason='needs the imported reference (build container only)')
def test_imported_reference_config_merges_the_dumped_cfg(cfgmod, tmp_path):
    """The `cfg` string of a detections.pkl written here goes through the REFERENCE's own
    load_cfg + merge_cfg_from_cfg (what tools/reval.py:88-93 does) in a child interpreter (the
    two packages share the name `detectron`): no unknown key, no type error, and the values
    that steer the reader come out as written."""
    import subprocess
    from detectron.core import test_engine_wsl as te
    _cfg(cfgmod, tmp_path, 2, 1)
    ab, asg, akp = te.empty_results(4, 2)
    f = str(tmp_path / 'detections.pkl')
    te.save_detections(f, ab, asg, akp)
    # the `cfg` entry of a weights file, as save_model_to_weights_file writes it
    import detectron.utils.env as envu
    from detectron.core.config import cfg
    from detectron.utils.net_wsl import save_object
    wf = str(tmp_path / 'model_final.pkl')
    save_object(dict(blobs={}, cfg=envu.yaml_dump(cfg, reference_format=True)), wf)
    script = textwrap.dedent('''
        import pickle, sys
        sys.dont_write_bytecode = True
        sys.path.insert(0, %r)
        import yaml
        from make_golden_from_reference import REF, _StubFinder
        sys.meta_path.insert(0, _StubFinder())
        sys.path.insert(0, REF)
        import future.utils
        future.utils.iteritems = lambda d: iter(d.items())
        import detectron.utils.env as envu
        # (the reference's `yaml.load` without a Loader argument is PyYAML's unsafe loader)
        envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.UnsafeLoader)
        from detectron.core import config as rcfg
        assert rcfg.__file__.startswith(REF)
        dets = pickle.load(open(%r, 'rb'))
        rcfg.merge_cfg_from_cfg(rcfg.load_cfg(dets['cfg']))
        c = rcfg.cfg
        print('MERGED', c.MODEL.NUM_CLASSES, c.TEST.BBOX_AUG.ENABLED, type(c.TEST.BBOX_AUG.SCALES).__name__,
              c.PIXEL_MEANS.shape, len(dets['all_boxes']), len(dets['all_segms'][1]))
        # a weights file written here, read the reference's way (utils/net_wsl.py:64-66, :277)
        saved = rcfg.load_cfg(pickle.load(open(%r, 'rb'))['cfg'])
        print('WEIGHTS_CFG', type(saved).__name__, 'MODEL' in saved, 'BBOX_REG_WEIGHTS' in saved.MODEL,
              saved.MODEL.TYPE, saved.WSL.DILATION)
    ''') % (os.path.join(ROOT, 'tests', 'golden'), f, wf)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env.pop('PYTHONPATH', None)
    r = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, env=env,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'MERGED 4 %s tuple (1, 1, 3) 4 2' % cfg.TEST.BBOX_AUG.ENABLED in r.stdout, r.stdout
    assert 'WEIGHTS_CFG AttrDict True True generalized_wsl 2' in r.stdout, r.stdout
