'use strict';
/*
 * Scene loading for Node: loadScene / loadOBJ / parseOBJ with the behaviour of
 * the reference's src/common.ts:34-148, reading files instead of fetching
 * URLs.  Plain JS that Node 12 parses (no optional chaining, no global fetch).
 */
const fs = require('fs');
const path = require('path');

// src/common.ts:3-10 without a DOM: the message goes to stderr.
function showError(message) {
    console.error('HIP Error:', message);
}

// src/common.ts:12-32
class RethrownError extends Error {
    constructor(message, error) {
        super(message);
        this.name = this.constructor.name;
        if (!error) throw new Error('RethrownError requires a message and error');
        this.original_error = error;
        this.stack_before_rethrow = this.stack;
        const messageLines = (this.message.match(/\n/g) || []).length + 1;
        this.stack =
            (this.stack ? this.stack.split('\n').slice(0, messageLines + 1).join('\n') : '') +
            '\n' + error.stack;
    }
}

// src/common.ts:45-109.  Quirks kept: split on '\n' and trim; faces split on a
// single space; a face vertex needs a position and a normal index (v//vn or
// v/vt/vn) or it is skipped; out-of-range indices are skipped with a warning;
// fan triangulation.
function parseOBJ(objText) {
    const triangles = [];
    const tempPositions = [];
    const tempNormals = [];
    const lines = objText.split('\n');
    for (let line of lines) {
        line = line.trim();
        if (line.startsWith('v ')) {
            const p = line.split(/\s+/);
            tempPositions.push([parseFloat(p[1]), parseFloat(p[2]), parseFloat(p[3])]);
        } else if (line.startsWith('vn ')) {
            const p = line.split(/\s+/);
            tempNormals.push([parseFloat(p[1]), parseFloat(p[2]), parseFloat(p[3])]);
        } else if (line.startsWith('f ')) {
            const parts = line.slice(2).trim().split(' ');
            if (parts.length < 3) continue;
            for (let i = 1; i < parts.length - 1; i++) {
                const triangleVertices = [];
                for (const fv of [parts[0], parts[i], parts[i + 1]]) {
                    const idx = fv.split('/').map((x) => (x ? parseInt(x) : undefined));
                    const vIdx = idx[0];
                    const nIdx = idx[2];
                    if (vIdx === undefined || nIdx === undefined) continue;
                    if (vIdx < 1 || vIdx > tempPositions.length || nIdx < 1 || nIdx > tempNormals.length) {
                        console.warn('Invalid index in face definition: ' + fv);
                        continue;
                    }
                    triangleVertices.push(tempPositions[vIdx - 1]);
                }
                triangles.push({
                    v0: triangleVertices[0],
                    v1: triangleVertices[1],
                    v2: triangleVertices[2],
                    color: [1, 1, 1],
                    emissionColor: [0, 0, 0],
                    emissionStrength: 0,
                    smoothness: 0,
                    specularProbability: 0,
                });
            }
        }
    }
    return { triangles: triangles };
}

// src/common.ts:34-43
async function loadOBJ(file) {
    const text = await fs.promises.readFile(file, 'utf8');
    const result = parseOBJ(text);
    console.log('Loaded OBJ from ' + file + ': ' + result.triangles.length + ' triangles');
    return result;
}

// src/common.ts:111-148.  Model paths resolve against the scene file's
// directory.  position / rotation / scale of a model are not applied, as in
// the reference; only the five material fields are copied.
async function loadScene(scenePath) {
    let text;
    try {
        text = await fs.promises.readFile(scenePath, 'utf8');
    } catch (e) {
        throw new Error('Failed to load scene file: ' + e.message);
    }
    const scene = JSON.parse(text);
    const base = path.dirname(path.resolve(scenePath));
    for (const model of scene.models || []) {
        if (model.path) {
            try {
                const objData = await loadOBJ(path.join(base, model.path));
                for (const triangle of objData.triangles) {
                    scene.triangles.push(Object.assign({}, triangle, {
                        color: model.color,
                        emissionColor: model.emissionColor,
                        emissionStrength: model.emissionStrength,
                        smoothness: model.smoothness,
                        specularProbability: model.specularProbability,
                    }));
                }
            } catch (error) {
                console.error('Failed to load model from ' + model.path + ':', error);
            }
        }
    }
    return { camera: scene.camera, spheres: scene.spheres, triangles: scene.triangles };
}

// src/common.ts:150-158
function formatNumber(num) {
    return new Intl.NumberFormat('en-US', { notation: 'compact', compactDisplay: 'short', maximumFractionDigits: 2 }).format(num);
}

module.exports = { showError, RethrownError, parseOBJ, loadOBJ, loadScene, formatNumber };
