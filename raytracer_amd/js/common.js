'use strict';
/*
 * Scene files for a Node host: loadScene / loadOBJ / parseOBJ / showError /
 * RethrownError / formatNumber, behaving as the functions of the same names in
 * the reference's src/common.ts:3-158, with files read from disk where the
 * reference fetches URLs.  Plain CommonJS that Node 12 parses.
 */
const fs = require('fs');
const path = require('path');
const util = require('util');

const readText = util.promisify(fs.readFile);

// The five fields a model hands to its triangles (src/common.ts:131-137); a model's position,
// rotation and scale are read and ignored there, so they are here.
const MATERIAL_KEYS = ['color', 'emissionColor', 'emissionStrength', 'smoothness', 'specularProbability'];

const DEFAULT_MATERIAL = {
    color: [1, 1, 1], emissionColor: [0, 0, 0], emissionStrength: 0, smoothness: 0, specularProbability: 0,
};

/* src/common.ts:3-10 shows the message in the page; a headless host writes it to stderr. */
function showError(message) {
    process.stderr.write('HIP Error: ' + message + '\n');
}

/*
 * An error that keeps the one it wraps (src/common.ts:12-32): `original_error`, the stack before
 * wrapping in `stack_before_rethrow`, and a stack made of this error's head (message lines + one
 * frame) followed by the wrapped error's stack.
 */
class RethrownError extends Error {
    constructor(message, error) {
        super(message);
        if (!error) throw new Error('RethrownError requires a message and error');
        this.name = 'RethrownError';
        this.original_error = error;
        this.stack_before_rethrow = this.stack;
        const own = String(this.stack || '').split('\n');
        const keep = String(message).split('\n').length + 1;
        this.stack = own.slice(0, keep).concat([error.stack]).join('\n');
    }
}

function numbersAfterKeyword(line) {
    const words = line.split(/\s+/);
    return [parseFloat(words[1]), parseFloat(words[2]), parseFloat(words[3])];
}

/*
 * One corner "v/vt/vn" of a face -> its position, or UNUSABLE when the corner cannot be used:
 * it needs a position index and a normal index (so "v" and "v/vt" are not enough), and both have
 * to point at something already read (1-based).
 */
const UNUSABLE = {};

function cornerPosition(corner, positions, normalCount) {
    const fields = corner.split('/');
    if (!fields[0] || !fields[2]) return UNUSABLE;
    const v = parseInt(fields[0], 10);
    const n = parseInt(fields[2], 10);
    if (v < 1 || v > positions.length || n < 1 || n > normalCount) {
        console.warn('Invalid index in face definition: ' + corner);
        return UNUSABLE;
    }
    return positions[v - 1]; // an index that is not a number passes the range test, as in the reference, and yields undefined
}

/*
 * Wavefront OBJ -> triangles (src/common.ts:45-109).  Lines are cut at "\n" and trimmed; "v" and
 * "vn" take three numbers; a face's corners are separated by exactly one space and a polygon
 * becomes a fan around its first corner.  Corners that cannot be used are left out, which leaves
 * that triangle with fewer than three vertices (v2, or all of them, undefined), as the reference
 * does; such a triangle fails later, when the scene is packed.
 */
function parseOBJ(objText) {
    const positions = [];
    let normalCount = 0;
    const triangles = [];
    objText.split('\n').forEach((raw) => {
        const line = raw.trim();
        if (line.startsWith('v ')) {
            positions.push(numbersAfterKeyword(line));
        } else if (line.startsWith('vn ')) {
            normalCount += 1;
        } else if (line.startsWith('f ')) {
            const corners = line.slice(2).trim().split(' ');
            for (let k = 2; k < corners.length; k++) {
                const found = [corners[0], corners[k - 1], corners[k]]
                    .map((c) => cornerPosition(c, positions, normalCount))
                    .filter((p) => p !== UNUSABLE);
                triangles.push(Object.assign({ v0: found[0], v1: found[1], v2: found[2] }, DEFAULT_MATERIAL));
            }
        }
    });
    return { triangles };
}

/* src/common.ts:34-43 with a file path in place of a URL. */
async function loadOBJ(file) {
    const parsed = parseOBJ(await readText(file, 'utf8'));
    console.log('Loaded OBJ from ' + file + ': ' + parsed.triangles.length + ' triangles');
    return parsed;
}

/*
 * A model vertex under the model's scale, rotation (degrees about x, then y, then z) and position,
 * in that order.  The reference declares these fields (src/types.ts:34-39) and never applies them;
 * this is an opt-in extension with a convention of its own.
 */
function transformPoint(p, model) {
    const s = model.scale || [1, 1, 1], r = model.rotation || [0, 0, 0], t = model.position || [0, 0, 0];
    const rad = (deg) => deg * Math.PI / 180;
    let x = p[0] * s[0], y = p[1] * s[1], z = p[2] * s[2];
    const cx = Math.cos(rad(r[0])), sx = Math.sin(rad(r[0]));
    [y, z] = [y * cx - z * sx, y * sx + z * cx];
    const cy = Math.cos(rad(r[1])), sy = Math.sin(rad(r[1]));
    [x, z] = [x * cy + z * sy, -x * sy + z * cy];
    const cz = Math.cos(rad(r[2])), sz = Math.sin(rad(r[2]));
    [x, y] = [x * cz - y * sz, x * sz + y * cz];
    return [x + t[0], y + t[1], z + t[2]];
}

/*
 * Scene JSON -> { camera, spheres, triangles } (src/common.ts:111-148).  Each model's OBJ file is
 * looked up next to the scene file (the reference serves both from public/), its triangles get the
 * model's material and are appended after the scene's own triangles, in model order.  A model
 * that cannot be loaded is reported and skipped.  options.applyModelTransforms (off by default: it
 * changes the image) sends the model's vertices through transformPoint.
 */
async function loadScene(scenePath, options) {
    let raw;
    try {
        raw = JSON.parse(await readText(scenePath, 'utf8'));
    } catch (e) {
        throw new Error('Failed to load scene file: ' + e.message);
    }
    const folder = path.dirname(path.resolve(scenePath));
    const triangles = raw.triangles;
    for (const model of raw.models || []) {
        if (!model.path) continue;
        try {
            const mesh = await loadOBJ(path.join(folder, model.path));
            mesh.triangles.forEach((t) => {
                const copy = Object.assign({}, t);
                if (options && options.applyModelTransforms) {
                    ['v0', 'v1', 'v2'].forEach((key) => { copy[key] = transformPoint(copy[key], model); });
                }
                MATERIAL_KEYS.forEach((key) => { copy[key] = model[key]; });
                triangles.push(copy);
            });
        } catch (error) {
            console.error('Failed to load model from ' + model.path + ':', error);
        }
    }
    return { camera: raw.camera, spheres: raw.spheres, triangles };
}

/* 1234567 -> "1.23M" (src/common.ts:150-158). */
function formatNumber(value) {
    const compact = new Intl.NumberFormat('en-US', { notation: 'compact', compactDisplay: 'short', maximumFractionDigits: 2 });
    return compact.format(value);
}

module.exports = { showError, RethrownError, parseOBJ, loadOBJ, loadScene, transformPoint, formatNumber };
