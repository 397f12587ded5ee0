'use strict';
/*
 * Byte layouts of the three GPU buffers, written by hand: the reference packs
 * them with typegpu from src/gpu-schemas.ts:3-38, which follows the WGSL
 * layout rules (vec3f aligned to 16 bytes).  Offsets are in floats.
 */
const SPHERE_FLOATS = 16;   // 64-byte stride
const TRIANGLE_FLOATS = 24; // 96-byte stride
const UNIFORMS_FLOATS = 20; // 80 bytes

function put3(dst, at, v) {
    dst[at] = v[0];
    dst[at + 1] = v[1];
    dst[at + 2] = v[2];
}

function packSpheres(spheres) {
    const out = new Float32Array(spheres.length * SPHERE_FLOATS);
    spheres.forEach((s, i) => {
        const o = i * SPHERE_FLOATS;
        put3(out, o + 0, s.center);
        out[o + 3] = s.radius;
        put3(out, o + 4, s.color);
        out[o + 7] = s.smoothness;
        put3(out, o + 8, s.emissionColor);
        out[o + 11] = s.emissionStrength;
        out[o + 12] = s.specularProbability;
    });
    return out;
}

function packTriangles(triangles) {
    const out = new Float32Array(triangles.length * TRIANGLE_FLOATS);
    triangles.forEach((t, i) => {
        const o = i * TRIANGLE_FLOATS;
        put3(out, o + 0, t.v0);
        put3(out, o + 4, t.v1);
        put3(out, o + 8, t.v2);
        put3(out, o + 12, t.color);
        put3(out, o + 16, t.emissionColor);
        out[o + 19] = t.emissionStrength;
        out[o + 20] = t.smoothness;
        out[o + 21] = t.specularProbability;
    });
    return out;
}

// frameIndex, samplesPerPixel and debugEnabled are f32, as src/webgpu-renderer.ts:217-229 writes them.
function packUniforms(camera, frameIndex, width, height, samplesPerPixel, debugEnabled) {
    const u = new Float32Array(UNIFORMS_FLOATS);
    put3(u, 0, camera.position);
    put3(u, 4, camera.rotation);
    u[7] = camera.fov;
    u[8] = camera.nearPlane;
    u[9] = camera.farPlane;
    u[12] = frameIndex;
    u[14] = width;
    u[15] = height;
    u[16] = samplesPerPixel;
    u[17] = debugEnabled ? 1 : 0;
    return u;
}

module.exports = { SPHERE_FLOATS, TRIANGLE_FLOATS, UNIFORMS_FLOATS, packSpheres, packTriangles, packUniforms };
