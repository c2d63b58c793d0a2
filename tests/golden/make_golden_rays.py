"""Mint tests/golden/rays_reference.npz by running the REFERENCE's own ray generation and position code (BUILD CONTAINER ONLY).

    python tests/golden/make_golden_rays.py

Two groups of vectors, each produced by reference classes only:
* ``rays/*``: one training batch of rays as the reference's data path forms it - pixel draws indices = floor(rand(n, 3) * [num_images, H, W])
  (data/pixel_samplers.py:45-48), the pixel centres image_coords[y, x] = (y + 0.5, x + 0.5) (model_components/ray_generators.py:40-63,
  cameras/cameras.py:276-302) and ``Cameras.generate_rays(camera_indices, coords)`` (cameras/cameras.py:304-696) for perspective cameras
  without distortion: origins, unit directions, directions_norm.  The product's sdfhip_generate_rays takes camera-to-world rotations with
  columns (x right, y DOWN, z FORWARD); nerfstudio's cameras look along -z with y up, so the reference is fed R_ref = R diag(1, -1, -1).
* ``pos/*``: frustum mid points ``Frustums.get_positions()`` and start points ``get_start_positions()`` (cameras/rays.py:46-73) followed
  by ``SceneContraction(order=inf | None)`` (field_components/spatial_distortions.py:42-92): what sdfhip_geo_forward_rays forms inside the
  encode kernel for the background field.
* ``image/*``: every pixel's ray of one camera (``Cameras.generate_rays(camera_indices=i)``, batch shape [H, W]) and one row-major chunk of it
  (``RayBundle.get_row_major_sliced_ray_bundle``): the inputs of ``Model.get_outputs_for_camera_ray_bundle`` (models/base_model.py:165-189).
Consumers: tests/test_gpu_glue.py (the kernels against these vectors), tests/test_cpu_oracle_and_abi.py (the fixture's own consistency)."""
import os
import sys

sys.dont_write_bytecode = True  # /root/reference is read-only: importing it must leave no __pycache__ there

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness  # noqa: E402


def reference_vectors():
    ref_harness.import_reference()
    from nerfstudio.cameras.cameras import Cameras, CameraType
    from nerfstudio.cameras.rays import Frustums
    from nerfstudio.field_components.spatial_distortions import SceneContraction

    out = {}
    gen = torch.Generator().manual_seed(17)
    C, H, W = 7, 48, 64
    fx, fy, cx, cy = 61.25, 60.5, 31.7, 23.9
    # camera poses: random rotations (columns x right, y down, z forward) and centres
    q = torch.linalg.qr(torch.randn(C, 3, 3, generator=gen))[0]
    q = q * torch.sign(torch.linalg.det(q))[:, None, None]
    centers = torch.randn(C, 3, generator=gen)
    c2w_ref = torch.cat([q * torch.tensor([1.0, -1.0, -1.0]), centers[:, :, None]], dim=-1)  # nerfstudio: x right, y up, z back
    cams = Cameras(camera_to_worlds=c2w_ref, fx=fx, fy=fy, cx=cx, cy=cy, height=H, width=W, camera_type=CameraType.PERSPECTIVE)
    n = 1024
    u = torch.rand(n, 3, generator=gen)
    u[0] = torch.tensor([0.0, 0.0, 0.0])
    u[1] = torch.tensor([0.999999, 0.999999, 0.999999])
    indices = torch.floor(u * torch.tensor([C, H, W])).long()  # pixel_samplers.py:45-48
    c, y, x = indices[:, 0], indices[:, 1], indices[:, 2]
    image_coords = cams.get_image_coords()  # [H, W, 2] = (y + 0.5, x + 0.5)
    coords = image_coords[y, x]  # ray_generators.py:56
    rb = cams.generate_rays(camera_indices=c[:, None], coords=coords)
    out.update({"rays/u": u.numpy(), "rays/rot": q.numpy(), "rays/centers": centers.numpy(),
                "rays/intrinsics": np.array([fx, fy, cx, cy, H, W, C], np.float64), "rays/indices": indices.numpy(),
                "rays/origins": rb.origins.numpy(), "rays/directions": rb.directions.numpy(), "rays/directions_norm": rb.directions_norm.numpy()})

    # frustum positions + contraction
    n, s = 96, 11
    o = torch.randn(n, 3, generator=gen) * 0.5
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    starts = torch.sort(torch.rand(n, s, generator=gen) ** 2 * 6.0, dim=-1).values  # near the origin (inside the unit ball) and far outside
    ends = starts + torch.rand(n, s, generator=gen) * 0.4 + 0.01
    fr = Frustums(origins=o[:, None, :].expand(n, s, 3), directions=d[:, None, :].expand(n, s, 3), starts=starts[..., None], ends=ends[..., None],
                  pixel_area=torch.ones(n, s, 1))
    out.update({"pos/origins": o.numpy(), "pos/directions": d.numpy(), "pos/starts": starts.numpy(), "pos/ends": ends.numpy()})
    for name, order in (("inf", float("inf")), ("l2", None)):
        con = SceneContraction(order=order)
        out[f"pos/mid_{name}"] = con(fr.get_positions()).numpy()
        out[f"pos/start_{name}"] = con(fr.get_start_positions()).numpy()
    mag = torch.linalg.norm(fr.get_positions(), ord=float("inf"), dim=-1)
    assert float((mag < 1).float().mean()) > 0.1 and float((mag > 1).float().mean()) > 0.3, "both branches of the contraction"

    # a whole camera image, as the eval path asks for it (Cameras.generate_rays(camera_indices=i): batch shape [H, W]), and one row-major
    # chunk of it as Model.get_outputs_for_camera_ray_bundle slices it (models/base_model.py:176-179, cameras/rays.py:282-293)
    cam_i = 3
    image = cams.generate_rays(camera_indices=cam_i)
    assert tuple(image.origins.shape) == (H, W, 3) and len(image) == H * W
    chunk = image.get_row_major_sliced_ray_bundle(100, 164)
    out.update({"image/camera": np.array(cam_i), "image/origins": image.origins.numpy(), "image/directions": image.directions.numpy(),
                "image/directions_norm": image.directions_norm.numpy(), "image/camera_indices": image.camera_indices.numpy(),
                "image/chunk_100_164_directions": chunk.directions.numpy(), "image/chunk_100_164_camera_indices": chunk.camera_indices.numpy()})
    return out


if __name__ == "__main__":
    vec = reference_vectors()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rays_reference.npz")
    np.savez_compressed(path, **vec)
    print("wrote", path, {k: v.shape for k, v in vec.items()})
