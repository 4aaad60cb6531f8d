"""A ctypes mirror of rtk_scene_desc (include/rtk.h) and a small builder, so that tests can hand the C ABI, the
optimiser and the oracle scenes that no named scene of scene_library.h covers (random soups, odd graphs)."""
import ctypes as C
import math


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class Node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("center0", Vec3), ("center_dir", Vec3), ("radius", C.c_double), ("material", C.c_int32), ("_pad", C.c_int32)]


class Quad(C.Structure):
    _fields_ = [("Q", Vec3), ("u", Vec3), ("v", Vec3), ("w", Vec3), ("normal", Vec3), ("D", C.c_double), ("material", C.c_int32), ("_pad", C.c_int32)]


class Triangle(C.Structure):
    _fields_ = [("p0", Vec3), ("p1", Vec3), ("p2", Vec3), ("normal", Vec3), ("uv0", C.c_float * 2), ("uv1", C.c_float * 2), ("uv2", C.c_float * 2),
                ("material", C.c_int32), ("_pad", C.c_int32)]


class Aabb(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")]


class Translate(C.Structure):
    _fields_ = [("offset", Vec3)]


class RotateY(C.Structure):
    _fields_ = [("sin_theta", C.c_double), ("cos_theta", C.c_double)]


class Medium(C.Structure):
    _fields_ = [("neg_inv_density", C.c_double), ("material", C.c_int32), ("_pad", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", Vec3), ("param", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("even", C.c_int32), ("odd", C.c_int32), ("image", C.c_int32), ("color", Vec3), ("param", C.c_double)]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("texel_offset", C.c_int64)]


class Perlin(C.Structure):
    _fields_ = [("randvec", (C.c_double * 3) * 256), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256), ("perm_z", C.c_int32 * 256)]


class PointLight(C.Structure):
    _fields_ = [("position", Vec3), ("intensity", Vec3), ("size", C.c_double)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("root", C.c_int32), ("n_nodes", C.c_int32), ("n_list_children", C.c_int32), ("n_spheres", C.c_int32),
                ("n_quads", C.c_int32), ("n_triangles", C.c_int32), ("n_bvh_boxes", C.c_int32), ("n_translates", C.c_int32), ("n_rotates", C.c_int32),
                ("n_media", C.c_int32), ("n_materials", C.c_int32), ("n_textures", C.c_int32), ("n_images", C.c_int32), ("n_perlins", C.c_int32),
                ("n_lights", C.c_int32), ("n_texel_bytes", C.c_int64),
                ("nodes", C.POINTER(Node)), ("list_children", C.POINTER(C.c_int32)), ("spheres", C.POINTER(Sphere)), ("quads", C.POINTER(Quad)),
                ("triangles", C.POINTER(Triangle)), ("bvh_boxes", C.POINTER(Aabb)), ("translates", C.POINTER(Translate)), ("rotates", C.POINTER(RotateY)),
                ("media", C.POINTER(Medium)), ("materials", C.POINTER(Material)), ("textures", C.POINTER(Texture)), ("images", C.c_void_p),
                ("texels", C.c_void_p), ("perlins", C.c_void_p), ("lights", C.c_void_p)]


NODE_SPHERE, NODE_QUAD, NODE_TRIANGLE, NODE_LIST, NODE_BVH, NODE_TRANSLATE, NODE_ROTATE_Y, NODE_MEDIUM = range(1, 9)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC, MAT_SPECULAR = range(1, 7)
TEX_SOLID, TEX_CHECKER, TEX_CHECKER_TRI, TEX_IMAGE, TEX_NOISE = range(1, 6)


def _union(boxes):
    return tuple((min if k % 2 == 0 else max)(b[k] for b in boxes) for k in range(6))


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


class DescBuilder:
    """Collects nodes and tables; `finish(root)` returns an object whose `.desc_ptr` is a valid rtk_scene_desc* (kept alive by it)."""

    def __init__(self):
        self.nodes, self.children, self.spheres, self.quads, self.translates, self.rotates, self.media = [], [], [], [], [], [], []
        self.triangles = []
        self.materials, self.textures = [], []
        self.images, self.texels, self.perlins, self.lights = [], bytearray(), [], []
        self.boxes = []       # rtk_aabb of the bvh nodes
        self.box_of = {}      # node -> (xmin, xmax, ymin, ymax, zmin, zmax) where the builder knows it (spheres and what is made of them)

    def _node(self, kind, a=0, b=0, c=0):
        self.nodes.append(Node(kind, a, b, c))
        return len(self.nodes) - 1

    def solid(self, rgb):
        self.textures.append(Texture(TEX_SOLID, 0, 0, 0, Vec3(*rgb), 0.0))
        return len(self.textures) - 1

    def checker(self, scale, even, odd, by_uv=False):
        """checker_texture (texture.h:34-56) over two textures, or the uv form (texture.h:58-84); param = 1 / scale."""
        self.textures.append(Texture(TEX_CHECKER_TRI if by_uv else TEX_CHECKER, even, odd, 0, Vec3(0, 0, 0), 1.0 / scale))
        return len(self.textures) - 1

    def noise(self, scale, rnd):
        """noise_texture (texture.h:110-120) over a perlin table filled the way perlin.h:6-13,59-70 fills it, from `rnd`."""
        pn = Perlin()
        for i in range(256):
            v = [rnd.uniform(-1, 1) for _ in range(3)]
            n = math.sqrt(sum(c * c for c in v)) or 1.0
            for k in range(3):
                pn.randvec[i][k] = v[k] / n
        for perm in (pn.perm_x, pn.perm_y, pn.perm_z):
            order = list(range(256))
            rnd.shuffle(order)
            for i in range(256):
                perm[i] = order[i]
        self.perlins.append(pn)
        self.textures.append(Texture(TEX_NOISE, 0, 0, len(self.perlins) - 1, Vec3(0, 0, 0), scale))
        return len(self.textures) - 1

    def image(self, width, height, rnd):
        """image_texture (texture.h:86-108) over random RGB8 texels (rtw_stb_image.h:71-81); width 0 = the missing-file colour."""
        self.images.append(Image(width, height, len(self.texels)))
        self.texels += bytes(rnd.randrange(256) for _ in range(width * height * 3))
        self.textures.append(Texture(TEX_IMAGE, 0, 0, len(self.images) - 1, Vec3(0, 0, 0), 0.0))
        return len(self.textures) - 1

    def textured(self, texture, kind=MAT_LAMBERTIAN):
        """lambertian / diffuse_light / isotropic over any texture."""
        self.materials.append(Material(kind, texture, Vec3(0, 0, 0), 0.0))
        return len(self.materials) - 1

    def specular(self, rgb, shininess):
        self.materials.append(Material(MAT_SPECULAR, -1, Vec3(*rgb), shininess))
        return len(self.materials) - 1

    def point_light(self, position, intensity, size):
        self.lights.append(PointLight(Vec3(*position), Vec3(*intensity), size))

    def lambertian(self, rgb):
        self.materials.append(Material(MAT_LAMBERTIAN, self.solid(rgb), Vec3(0, 0, 0), 0.0))
        return len(self.materials) - 1

    def metal(self, rgb, fuzz):
        self.materials.append(Material(MAT_METAL, -1, Vec3(*rgb), min(fuzz, 1.0)))
        return len(self.materials) - 1

    def dielectric(self, index):
        self.materials.append(Material(MAT_DIELECTRIC, -1, Vec3(0, 0, 0), index))
        return len(self.materials) - 1

    def light(self, rgb):
        self.materials.append(Material(MAT_DIFFUSE_LIGHT, self.solid(rgb), Vec3(0, 0, 0), 0.0))
        return len(self.materials) - 1

    def sphere(self, centre, radius, material, motion=(0.0, 0.0, 0.0)):
        self.spheres.append(Sphere(Vec3(*centre), Vec3(*motion), max(0.0, radius), material, 0))
        node = self._node(NODE_SPHERE, len(self.spheres) - 1)
        r = max(0.0, radius)
        ends = [centre, tuple(centre[k] + motion[k] for k in range(3))]                     # sphere.h:17,24-26
        self.box_of[node] = tuple(f(e[k] + sgn * r for e in ends) for k in range(3) for f, sgn in ((min, -1), (max, 1)))
        return node

    def quad(self, Q, u, v, material):
        n = _cross(u, v)                                   # quad.h:12-19
        nn = _dot(n, n)
        length = math.sqrt(nn)
        normal = tuple((1 / length) * c for c in n)        # unit_vector: v / length = (1/length) * v (vec3.h:91-93,104-106)
        w = tuple((1 / nn) * c for c in n)                 # n / dot(n, n)
        self.quads.append(Quad(Vec3(*Q), Vec3(*u), Vec3(*v), Vec3(*w), Vec3(*normal), _dot(normal, Q), material, 0))
        node = self._node(NODE_QUAD, len(self.quads) - 1)
        corners = [Q, tuple(Q[k] + u[k] for k in range(3)), tuple(Q[k] + v[k] for k in range(3)), tuple(Q[k] + u[k] + v[k] for k in range(3))]
        self.box_of[node] = self._padded(corners)            # a bounding box of the four corners (quad.h:21-26 pads thin boxes as well)
        return node

    @staticmethod
    def _padded(points, pad=1e-4):
        return tuple(f(p[k] for p in points) + sgn * pad for k in range(3) for f, sgn in ((min, -1), (max, 1)))

    def triangle(self, p0, p1, p2, material, uvs=((0.0, 0.0), (1.0, 0.0), (0.0, 1.0))):
        e1 = tuple(p1[k] - p0[k] for k in range(3))          # triangle.h:21-23: n = cross(p1 - p0, p2 - p0), normal = unit_vector(n)
        e2 = tuple(p2[k] - p0[k] for k in range(3))
        n = _cross(e1, e2)
        length = math.sqrt(_dot(n, n))
        inv = 1 / length if length > 0 else math.inf        # a zero-area triangle: (1/0) * 0 = NaN components, as in C++
        normal = tuple(inv * c for c in n)
        f2 = C.c_float * 2
        self.triangles.append(Triangle(Vec3(*p0), Vec3(*p1), Vec3(*p2), Vec3(*normal), f2(*uvs[0]), f2(*uvs[1]), f2(*uvs[2]), material, 0))
        node = self._node(NODE_TRIANGLE, len(self.triangles) - 1)
        self.box_of[node] = self._padded([p0, p1, p2])
        return node

    def rank(self, node, rank):
        """rtk_node.c of a primitive node: 1 + its rank in the reference's visiting order (what rtk_scene_optimize records)."""
        self.nodes[node].c = rank
        return node

    def list(self, members):
        first = len(self.children)
        self.children.extend(members)
        node = self._node(NODE_LIST, first, len(members))
        if all(m in self.box_of for m in members) and members:
            self.box_of[node] = _union([self.box_of[m] for m in members])
        return node

    def isotropic(self, rgb):
        self.materials.append(Material(MAT_ISOTROPIC, self.solid(rgb), Vec3(0, 0, 0), 0.0))
        return len(self.materials) - 1

    def medium(self, boundary, density, rgb):
        """constant_medium(boundary, density, albedo) (constant_medium.h:12-17): neg_inv_density = -1 / density, isotropic phase."""
        self.media.append(Medium(-1.0 / density, self.isotropic(rgb), 0))
        node = self._node(NODE_MEDIUM, len(self.media) - 1, boundary)
        if boundary in self.box_of:
            self.box_of[node] = self.box_of[boundary]          # constant_medium.h:55
        return node

    def bvh(self, members, rnd):
        """bvh_node(list) over nodes with known boxes (spheres, media bounded by them, lists and bvh nodes of those): a random
        axis per node, sorted by box minimum, split at the middle; a span of one stores the object twice (bvh.h:13-45 in
        the book's form -- the test only needs A valid hierarchy the optimiser has to respect, boxes bounding their content)."""
        members = list(members)
        axis = rnd.randrange(3)
        members.sort(key=lambda m: self.box_of[m][2 * axis])
        if len(members) == 1:
            left = right = members[0]
        elif len(members) == 2:
            left, right = members
        else:
            mid = len(members) // 2
            left, right = self.bvh(members[:mid], rnd), self.bvh(members[mid:], rnd)
        box = _union([self.box_of[left], self.box_of[right]])
        self.boxes.append(Aabb(*box))
        node = self._node(NODE_BVH, left, right, len(self.boxes) - 1)
        self.box_of[node] = box
        return node

    def translate(self, child, offset):
        self.translates.append(Translate(Vec3(*offset)))
        return self._node(NODE_TRANSLATE, len(self.translates) - 1, child)

    def rotate_y(self, child, degrees):
        r = degrees * 3.1415926535897932385 / 180.0        # rtweekend.h:21-23
        self.rotates.append(RotateY(math.sin(r), math.cos(r)))
        return self._node(NODE_ROTATE_Y, len(self.rotates) - 1, child)

    def finish(self, root):
        return BuiltDesc(self, root)


class BuiltDesc:
    def __init__(self, b, root):
        def arr(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            return a
        self._keep = dict(nodes=arr(Node, b.nodes), children=(C.c_int32 * max(1, len(b.children)))(*b.children), spheres=arr(Sphere, b.spheres),
                          quads=arr(Quad, b.quads), translates=arr(Translate, b.translates), rotates=arr(RotateY, b.rotates), media=arr(Medium, b.media),
                          materials=arr(Material, b.materials), textures=arr(Texture, b.textures), tris=arr(Triangle, b.triangles), boxes=arr(Aabb, b.boxes))
        self._keep.update(images=arr(Image, b.images), texels=(C.c_uint8 * max(1, len(b.texels))).from_buffer_copy(bytes(b.texels) or b"\0"),
                          perlins=arr(Perlin, b.perlins), lights=arr(PointLight, b.lights))
        k = self._keep
        d = SceneDesc()
        d.abi_version, d.root = 2, root  # RTK_ABI_VERSION
        d.n_nodes, d.n_list_children, d.n_spheres, d.n_quads = len(b.nodes), len(b.children), len(b.spheres), len(b.quads)
        d.n_triangles = len(b.triangles)
        d.n_bvh_boxes = len(b.boxes)
        d.n_translates, d.n_rotates, d.n_media, d.n_materials, d.n_textures = len(b.translates), len(b.rotates), len(b.media), len(b.materials), len(b.textures)
        d.nodes, d.list_children, d.spheres, d.quads = k["nodes"], k["children"], k["spheres"], k["quads"]
        d.triangles, d.bvh_boxes, d.translates, d.rotates = k["tris"], k["boxes"], k["translates"], k["rotates"]
        d.media, d.materials, d.textures = k["media"], k["materials"], k["textures"]
        d.n_images, d.n_perlins, d.n_lights, d.n_texel_bytes = len(b.images), len(b.perlins), len(b.lights), len(b.texels)
        d.images, d.texels = C.cast(k["images"], C.c_void_p), C.cast(k["texels"], C.c_void_p)
        d.perlins, d.lights = C.cast(k["perlins"], C.c_void_p), C.cast(k["lights"], C.c_void_p)
        self.desc = d
        self.name = "built"

    @property
    def desc_ptr(self):
        return C.addressof(self.desc)
